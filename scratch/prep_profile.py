"""Host-side cost of turning the reference's holders into CSR tables + GRU schedules (CPU only)."""
import sys, os, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'jt-vae-tmcinvdes_amd'))
import numpy as np, torch
from fast_jtnn_hip import synthetic, tensorize as T, graphprep as G
trees = synthetic.make_batch(1234, 1024, vocab_size=385, mean_atoms=40.0)
t0 = time.perf_counter(); jt, _ = T.jtenc_tensorize(trees); mp = T.mpn_tensorize([t.mol for t in trees]); t1 = time.perf_counter()
print("tensorize (reference-format holders): %.1f ms" % ((t1 - t0) * 1e3))
dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
def prep():
    P = G.prepare_mpn(mp, dev)
    Pt = G.prepare_tree(jt, dev, 385)
    s = G.tree_schedule(Pt, 20)
    b = G.backward_schedule(Pt, s, False)
    return P, Pt, s, b
prep()
for name, fn in (("prepare_mpn", lambda: G.prepare_mpn(mp, dev)), ("prepare_tree", lambda: G.prepare_tree(jt, dev, 385))):
    t0 = time.perf_counter()
    for _ in range(5): fn()
    print("%s: %.1f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
Pt = G.prepare_tree(jt, dev, 385)
t0 = time.perf_counter()
for _ in range(5):
    Pt = G.prepare_tree(jt, dev, 385); s = G.tree_schedule(Pt, 20)
t1 = time.perf_counter()
for _ in range(5):
    Pt = G.prepare_tree(jt, dev, 385); s = G.tree_schedule(Pt, 20); b = G.backward_schedule(Pt, s, False)
t2 = time.perf_counter()
print("prepare_tree+tree_schedule: %.1f ms; + backward_schedule: %.1f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): prep()
if dev.type == 'cuda': torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
