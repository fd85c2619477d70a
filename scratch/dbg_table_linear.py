import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jt-vae-tmcinvdes_amd"))
import torch
from fast_jtnn_hip.linear import table_linear
from fast_jtnn_hip import _capi as K
n, nh, m, ns, out_f = 12001, 450, 385, 450, 450
gen = torch.Generator(device="cpu").manual_seed(n + 7 * m)
lin = torch.nn.Linear(nh + ns, out_f)
with torch.no_grad():
    lin.weight.copy_(0.2 * torch.randn(lin.weight.shape, generator=gen))
    lin.bias.copy_(0.2 * torch.randn(lin.bias.shape, generator=gen))
h, src = torch.randn((n, nh), generator=gen), torch.randn((m, ns), generator=gen)
idx = torch.randint(0, m - 1, (n,), generator=gen)
g = torch.randn((n, out_f), generator=gen)
for split in (True, False):
    K.set_split_products(split)
    mod = torch.nn.Linear(nh + ns, out_f).cuda(); mod.load_state_dict(lin.state_dict())
    hh = h.cuda().requires_grad_(True); ss = src.cuda().requires_grad_(True)
    y = table_linear(hh, ss, idx.cuda().int(), mod, h_first=False, relu=True)
    (y * g.cuda()).sum().backward()
    mod64 = torch.nn.Linear(nh + ns, out_f).double(); mod64.load_state_dict(lin.state_dict())
    h64 = h.double().requires_grad_(True); s64 = src.double().requires_grad_(True)
    y64 = torch.relu(mod64(torch.cat([s64[idx], h64], 1)))
    (y64 * g.double()).sum().backward()
    err = (hh.grad.double().cpu() - h64.grad).abs()
    bad = (err > 1e-3).nonzero()
    print("split", split, "max err", float(err.max()), "bad entries", bad.shape[0], "rows", bad[:, 0].unique()[:10].tolist(), "n bad rows", bad[:,0].unique().numel(), "cols", bad[:, 1].unique()[:12].tolist())
    # is it the relu mask?
    ydiff = ((y.detach().cpu() > 0) != (y64 > 0)).nonzero()
    print("   mask flips", ydiff.shape[0], ydiff[:5].tolist())
