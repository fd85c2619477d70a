"""Experiment: cProfile of holder preparation (one thread) on the GPU box."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import synthetic, tensorize as T
from fast_jtnn_hip.prefetch import warm_encoder_holders
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
train = (sys.argv[2] if len(sys.argv) > 2 else "train") == "train"
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, "train")
bs = []
for i in range(9):
    trees = synthetic.make_batch(3 + i, B, vocab_size=385, mean_atoms=40.0)
    bs.append((T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])))
for b in bs[:3]: warm_encoder_holders(model, b[0], b[1], train=train)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in bs[3:6]: warm_encoder_holders(model, b[0], b[1], train=train)
torch.cuda.synchronize()
print("ms per batch (plain)", 1e3 * (time.perf_counter() - t0) / 3)
pr = cProfile.Profile(); pr.enable()
for b in bs[6:]: warm_encoder_holders(model, b[0], b[1], train=train)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
