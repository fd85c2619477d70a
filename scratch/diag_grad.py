import sys, numpy as np, torch
sys.path.insert(0, 'jt-vae-tmcinvdes_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from common import fro_err, rel_err, golden_loss, make_hip_model, make_oracle
from fast_jtnn_hip import synthetic, tensorize as T
class G: pass
g = G(); g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = 21, 32, 450, 56, 20, 3, 385, 0.05
trees = synthetic.make_batch(21, 32, vocab_size=g.V, mean_atoms=40.0, with_cands=True)
jt, md = T.jtenc_tensorize(trees); mp = T.mpn_tensorize([t.mol for t in trees])
cands, _ = synthetic.collect_candidates(trees); cj = T.jtmpn_tensorize(cands, md)
hip = make_hip_model(g, 'cuda'); o32 = make_oracle(g); o64 = make_oracle(g).double()
def run(model, dbl=False):
    jt_, mp_, cj_ = jt, mp, cj
    if dbl:
        mp_ = (mp[0].double(), mp[1].double()) + tuple(mp[2:]); cj_ = (cj[0].double(), cj[1].double()) + tuple(cj[2:])
    tv, ms, mv = model.encode(jt_, mp_); cv = model.jtmpn(*cj_, ms)
    dev = tv.device
    mean = torch.cat([model.T_mean(tv), model.G_mean(mv)], 1); lv = torch.cat([-torch.abs(model.T_var(tv)), -torch.abs(model.G_var(mv))], 1)
    from common import projection
    loss = ((mean * projection(mean.shape, g.seed+101).to(dev).to(mean.dtype)).sum() + (lv * projection(lv.shape, g.seed+102).to(dev).to(mean.dtype)).sum()
            + (cv * projection(cv.shape, g.seed+103).to(dev).to(mean.dtype)).sum() / cv.shape[0] * g.B)
    loss.backward(); return loss
la, lb = run(hip), run(o32)
torch.set_default_dtype(torch.float64)
lc = run(o64, True)
torch.set_default_dtype(torch.float32)
print('loss hip %.8f o32 %.8f o64 %.8f' % (la.item(), lb.item(), lc.item()))
p32, p64 = dict(o32.named_parameters()), dict(o64.named_parameters())
for n, p in hip.named_parameters():
    if n not in p64 or p64[n].grad is None: continue
    r = p64[n].grad.numpy()
    print('%-26s hip-vs-64 fro %.2e max %.2e | cpu32-vs-64 fro %.2e max %.2e' % (n, fro_err(p.grad.cpu().numpy(), r), rel_err(p.grad.cpu().numpy(), r), fro_err(p32[n].grad.numpy(), r), rel_err(p32[n].grad.numpy(), r)))
