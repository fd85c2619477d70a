"""Full batch vs mean of its halves (the linearity check of tests/test_gpu_training_side.py), every gradient's
relative Frobenius deviation printed: run with JTVAE_SPLIT_BF16=0 / 1 to compare the two product forms."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_training_side import make_hip_model
from fast_jtnn_hip import synthetic, tensorize as T

B, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 385
trees = synthetic.make_batch(4242, B, vocab_size=V, mean_atoms=40.0, with_cands=True)

def x_batch(part):
    synthetic.set_batch_nodeID(part)
    jt, mess_dict = T.jtenc_tensorize(part)
    mp = T.mpn_tensorize([t.mol for t in part])
    cands, bi = synthetic.collect_candidates(part)
    return part, jt, mp, (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(bi))

class G: pass
g = G()
g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = 7, B, 450, 56, 20, 3, V, 0.05
m = make_hip_model(g, "cuda")
rs = np.random.RandomState(5)
eps = [torch.from_numpy(rs.standard_normal((B, 28)).astype(np.float32)).cuda() for _ in range(2)]

def run(lo, hi):
    m.zero_grad(set_to_none=True)
    out = m(x_batch(trees[lo:hi]), 0.005, epsilons=[e[lo:hi] for e in eps])
    out[0].backward(); torch.cuda.synchronize()
    return float(out[0].item()), {n: p.grad.detach().clone() for n, p in m.named_parameters()}

full, g_full = run(0, B)
a, g_a = run(0, B // 2)
b, g_b = run(B // 2, B)
print("split form:", os.environ.get("JTVAE_SPLIT_BF16", "1"), " loss full %.7f  mean of halves %.7f" % (full, 0.5 * (a + b)))
for name, gf in g_full.items():
    mean = 0.5 * (g_a[name] + g_b[name])
    print("  %-28s %.3e" % (name, float((gf - mean).norm() / mean.norm().clamp_min(1e-30))))
