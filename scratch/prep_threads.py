"""Experiment: does holder preparation scale over threads (how much of it holds the interpreter lock)?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import numpy as np, torch
from fast_jtnn_hip import graphprep, synthetic, tensorize as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = sys.argv[2] if len(sys.argv) > 2 else "cuda:0"
trees = synthetic.make_batch(3, B, vocab_size=385, mean_atoms=40.0)
jt, mp = T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])
def prep():
    p = graphprep.prepare_tree(jt, dev, 385)
    s = graphprep.tree_schedule(p, 20, mode="once")
    graphprep.backward_schedule(p, s, all_messages=False)
    graphprep.prepare_mpn(mp, dev, 0)
def run(nthreads, reps):
    def work():
        for _ in range(reps): prep()
    ts = [threading.Thread(target=work) for _ in range(nthreads)]
    if dev != "cpu": torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    if dev != "cpu": torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (nthreads * reps)
prep(); prep()
for n in (1, 2, 4):
    print("threads %d: %.2f ms per prepared batch" % (n, 1e3 * run(n, 4)))
