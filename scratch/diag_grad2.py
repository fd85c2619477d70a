import sys, numpy as np, torch
sys.path.insert(0, 'jt-vae-tmcinvdes_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from common import fro_err, rel_err, make_hip_model, make_oracle, projection
from fast_jtnn_hip import synthetic, tensorize as T
class G: pass
g = G(); g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = 21, 32, 450, 56, 20, 3, 385, 0.05
trees = synthetic.make_batch(21, 32, vocab_size=g.V, mean_atoms=40.0, with_cands=True)
jt, md = T.jtenc_tensorize(trees); mp = T.mpn_tensorize([t.mol for t in trees])
cands, _ = synthetic.collect_candidates(trees); cj = T.jtmpn_tensorize(cands, md)
hip = make_hip_model(g, 'cuda'); o32 = make_oracle(g); o64 = make_oracle(g).double()
with torch.no_grad():
    tv_h, ms_h = hip.jtnn(*jt); tv_32, ms_32 = o32.jtnn(*jt)
    torch.set_default_dtype(torch.float64); tv_64, ms_64 = o64.jtnn(*jt); torch.set_default_dtype(torch.float32)
print('messages: hip-vs-64 max %.2e fro %.2e | cpu32-vs-64 max %.2e fro %.2e' % (rel_err(ms_h.cpu().numpy(), ms_64.numpy()), fro_err(ms_h.cpu().numpy(), ms_64.numpy()), rel_err(ms_32.numpy(), ms_64.numpy()), fro_err(ms_32.numpy(), ms_64.numpy())))
print('tree_vecs: hip %.2e cpu32 %.2e' % (rel_err(tv_h.cpu().numpy(), tv_64.numpy()), rel_err(tv_32.numpy(), tv_64.numpy())))
# JTMPN alone on identical inputs
base = ms_64.float()
R = None
res = {}
for name, model, dev, dbl in (('hip', hip, 'cuda', False), ('o32', o32, 'cpu', False), ('o64', o64, 'cpu', True)):
    tm = base.to(dev).clone()
    cjj = cj
    if dbl:
        tm = tm.double(); cjj = (cj[0].double(), cj[1].double()) + tuple(cj[2:])
    tm.requires_grad_(True)
    model.zero_grad()
    cv = model.jtmpn(*cjj, tm)
    if R is None: R = projection(cv.shape, 7)
    (cv * R.to(dev).to(cv.dtype)).sum().backward()
    res[name] = dict(cv=cv.detach().cpu().numpy(), dtm=tm.grad.cpu().numpy()[1:], **{n: p.grad.cpu().numpy() for n, p in model.jtmpn.named_parameters()})
for k in res['o64']:
    r = res['o64'][k]
    print('%-12s hip fro %.2e max %.2e | cpu32 fro %.2e max %.2e' % (k, fro_err(res['hip'][k], r), rel_err(res['hip'][k], r), fro_err(res['o32'][k], r), rel_err(res['o32'][k], r)))

# tree encoder alone with identical upstream gradients
R1, R2 = projection(tv_64.shape, 1), projection(ms_64.shape, 2) * 0.01
res = {}
for name, model, dev, dbl in (('hip', hip, 'cuda', False), ('o32', o32, 'cpu', False), ('o64', o64, 'cpu', True)):
    model.zero_grad()
    if dbl: torch.set_default_dtype(torch.float64)
    tv, ms = model.jtnn(*jt)
    ((tv * R1.to(dev).to(tv.dtype)).sum() + (ms * R2.to(dev).to(tv.dtype)).sum()).backward()
    torch.set_default_dtype(torch.float32)
    res[name] = {n: p.grad.cpu().numpy() for n, p in model.jtnn.named_parameters()}
for k in res['o64']:
    r = res['o64'][k]
    print('%-22s hip fro %.2e max %.2e | cpu32 fro %.2e max %.2e' % (k, fro_err(res['hip'][k], r), rel_err(res['hip'][k], r), fro_err(res['o32'][k], r), rel_err(res['o32'][k], r)))
