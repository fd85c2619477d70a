"""Where the host time of flatprep.encoder_batch goes (cProfile + stage timers).  python scratch/flat_profile.py [batch] [train]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jt-vae-tmcinvdes_amd")]
import numpy as np, torch
from fast_jtnn_hip import JTNNVAE, flat, flatprep, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
train = len(sys.argv) > 2 and sys.argv[2] == "train"
V = 385
trees = synthetic.make_batch(1, 2 * B, vocab_size=V)
shard = flat.shard_from_trees(trees)
m = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3, decoder=False).cuda()
ids = [np.arange(0, B), np.arange(B, 2 * B)]
for th in (1, 4, 8, 16):
    for _ in range(2):
        flatprep.encoder_batch(m, shard, ids[0], train=train, threads=th)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        flatprep.encoder_batch(m, shard, ids[i & 1], train=train, threads=th)
    print("threads %2d: %.2f ms per batch (host)" % (th, (time.perf_counter() - t0) / 6 * 1e3))
    torch.cuda.synchronize()
for th in (1, 8):
    t0 = time.perf_counter()
    for i in range(6):
        flat.graph_csr(shard.mols, ids[i & 1], train=train, threads=th)
    t1 = time.perf_counter()
    for i in range(6):
        flat.tree_csr(shard.trees, ids[i & 1], threads=th)
    t2 = time.perf_counter()
    print("threads %d: graph_csr %.2f ms, tree_csr %.2f ms (numpy memory)" % (th, (t1 - t0) / 6e-3, (t2 - t1) / 6e-3))
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    flatprep.encoder_batch(m, shard, ids[i & 1], train=train, threads=8)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
