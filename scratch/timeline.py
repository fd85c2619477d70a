"""Experiment: timeline of one training step with the two encoder streams overlapping, from HIP events
around every C-ABI launch (the profiler's tracing slows the host enough to change the picture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import _capi as K, parallel
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
base = torch.cuda.Event(enable_timing=True)
with K.KernelTimer(streams=True) as kt:
    for _ in range(6): step()
    base.record()
    step()
    end = torch.cuda.Event(enable_timing=True); end.record()
torch.cuda.synchronize()
n = len(kt.records) // 7
recs = kt.records[-n:]; strs = kt.streams[-n:]
ids = {s: i for i, s in enumerate(dict.fromkeys(strs))}
print("step: %.3f ms (base -> end), %d launches" % (base.elapsed_time(end), n))
for (name, nb, fl, e0, e1, tag, _n), s in zip(recs, strs):
    print("%9.1f %8.1f s%d %-18s %s" % (1e3 * base.elapsed_time(e0), 1e3 * e0.elapsed_time(e1), ids[s], name, tag))
