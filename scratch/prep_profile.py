"""scratch/prep_profile.py [B] [train|encode]: host time of one batch drawn from flat storage (tensorize + preparation +
copies), profiled in the calling thread."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jt-vae-tmcinvdes_amd"))
from fast_jtnn_hip import JTNNVAE, flat, flatprep, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
train = (sys.argv[2] if len(sys.argv) > 2 else "train") == "train"
k = 6
trees = [t for i in range(k) for t in synthetic.make_batch(900 + 13 * i, B, vocab_size=385, mean_atoms=40.0)]
synthetic.set_batch_nodeID(trees)
shard = flat.shard_from_trees(trees)
model = JTNNVAE(synthetic.FakeVocab(385), 450, 56, 20, 3, decoder=False).cuda()
ids = [np.arange(i * B, (i + 1) * B) for i in range(k)]
threads = flat.host_threads()
build = lambda sel: flatprep.encoder_batch(model, shard, sel, train=train, threads=threads)  # noqa: E731
for sel in ids[:2]:
    build(sel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(5):
    for sel in ids:
        build(sel)
dt = (time.perf_counter() - t0) / (5 * k)
torch.cuda.synchronize()
print("B=%d train=%s threads=%d: %.3f ms of host time per batch" % (B, train, threads, 1e3 * dt))
pr = cProfile.Profile()
pr.enable()
for r in range(5):
    for sel in ids:
        build(sel)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
