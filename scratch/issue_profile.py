"""Experiment: host time of issuing one encoder training step (cProfile of the stepping thread)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import parallel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, "train")
model.overlap_streams = True
bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])
def step():
    bucket.zero()
    mean, logvar = model.encode_latent(jt, mp)
    (mean.sum() + logvar.sum()).backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("issue %.3f ms/step, total %.3f" % (1e3 * (t1 - t0) / 20, 1e3 * (time.perf_counter() - t0) / 20))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25); pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
