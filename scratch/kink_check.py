"""Experiment: how far is the reference's own fp32 gradient (golden) from the same op sequence in fp64?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from common import Golden, golden_cases, golden_x_batch, make_oracle_vae, rel_err
for path in golden_cases()[1:2]:
    g = Golden(path); z = g.z
    m = make_oracle_vae(g).double()
    trees, jt, mp, (cj, bi) = golden_x_batch(g)
    mp = (mp[0].double(), mp[1].double()) + tuple(mp[2:])
    cj = (cj[0].double(), cj[1].double()) + tuple(cj[2:])
    eps = (torch.from_numpy(z["fwd_eps_tree"]).double(), torch.from_numpy(z["fwd_eps_mol"]).double())
    loss = m((trees, jt, mp, (cj, bi)), float(z["fwd_beta"]), epsilons=eps)[0]
    loss.backward()
    print(g.name, "fp64 loss", loss.item(), "reference fp32", float(z["fwd_loss"]))
    for name, p in m.named_parameters():
        h = z["fghead/" + name]
        if np.abs(h).max() < 1e-8: continue
        e = rel_err(p.grad.reshape(-1)[:256].numpy(), h)
        if e > 1e-4:
            print("   %-28s reference-fp32 vs fp64 head rel %.2e" % (name, e))
