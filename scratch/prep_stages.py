"""Experiment: wall time of the stages of holder preparation on the GPU box (synchronised after each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import numpy as np, torch
from fast_jtnn_hip import graphprep, synthetic, tensorize as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
def fresh():
    trees = synthetic.make_batch(np.random.randint(1 << 20), B, vocab_size=385, mean_atoms=40.0)
    return T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])
def timed(what, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-28s %7.2f ms" % (what, 1e3 * (time.perf_counter() - t0))); return r
for rep in range(2):
    jt, mp = fresh()
    print("--- rep", rep, "B", B)
    p = timed("prepare_tree", lambda: graphprep.prepare_tree(jt, dev, 385))
    s = timed("tree_schedule", lambda: graphprep.tree_schedule(p, 20, mode="once"))
    timed("backward_schedule(roots)", lambda: graphprep.backward_schedule(p, s, all_messages=False))
    fa, fb = mp[0].numpy(), mp[1].numpy()
    timed("features pageable->dev", lambda: (graphprep._upload_features(fa, dev), graphprep._upload_features(fb, dev)))
    timed("features u8 convert host", lambda: (fa.astype(np.uint8), fb.astype(np.uint8)))
    pin = torch.empty(fb.shape, dtype=torch.float32).pin_memory()
    timed("fbonds memcpy to pinned", lambda: pin.numpy().__setitem__(slice(None), fb))
    timed("fbonds pinned->dev", lambda: pin.to(dev, non_blocking=True))
    z = (mp[0][:0], mp[1][:0]) 
    timed("prepare_mpn (all)", lambda: graphprep.prepare_mpn(mp, dev, 0))
    timed("  padded_to_csr x2", lambda: (graphprep.padded_to_csr(mp[3].numpy(), 1 << 30), graphprep.padded_to_csr(mp[2].numpy(), 1 << 30)))
