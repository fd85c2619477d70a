"""Experiment: host cost of holder preparation (device = cpu: index work only, no uploads)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
from fast_jtnn_hip import graphprep, synthetic, tensorize as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
train = len(sys.argv) > 2 and sys.argv[2] == "train"
trees = synthetic.make_batch(5, B, vocab_size=385, mean_atoms=40.0)
t0 = time.perf_counter(); jt = T.jtenc_tensorize(trees)[0]; mp = T.mpn_tensorize([t.mol for t in trees]); t1 = time.perf_counter()
print("tensorize %.1f ms" % (1e3 * (t1 - t0)))
def prep():
    p = graphprep.prepare_tree(jt, "cpu", 385)
    s = graphprep.tree_schedule(p, 20, mode="once")
    if train:
        graphprep.backward_schedule(p, s, all_messages=False)
    graphprep.prepare_mpn(mp, "cpu", 0)
prep()
t0 = time.perf_counter(); prep(); print("prep %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); prep(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
