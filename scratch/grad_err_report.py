"""Experiment: per-parameter gradient error of JTNNVAE.forward (HIP) against the golden's statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from common import Golden, golden_cases, golden_x_batch, make_hip_model, rel_err
for path in golden_cases():
    g = Golden(path); z = g.z
    m = make_hip_model(g, "cuda")
    eps = (torch.from_numpy(z["fwd_eps_tree"]).cuda(), torch.from_numpy(z["fwd_eps_mol"]).cuda())
    loss = m(golden_x_batch(g), float(z["fwd_beta"]), epsilons=eps)[0]
    loss.backward()
    print(g.name, "loss", loss.item(), float(z["fwd_loss"]))
    for name, p in m.named_parameters():
        k = "fgnorm/" + name
        if k not in z.files: continue
        ref = float(z[k]); got = float(p.grad.double().norm())
        h = z["fghead/" + name]
        e = rel_err(p.grad.reshape(-1)[:256].cpu().numpy(), h) if np.abs(h).max() > 1e-8 else 0.0
        if e > 1e-4 or abs(got - ref) > 1e-4 * ref:
            print("   %-28s norm rel %.2e head rel %.2e" % (name, abs(got - ref) / max(ref, 1e-9), e))
