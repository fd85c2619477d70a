"""Experiment: host profile of encoder steps on holders never seen before (index preparation inside the step)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import parallel, synthetic, tensorize as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, "train")
bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])
def step(jt_, mp_):
    if mode == "encode":
        with torch.no_grad():
            model.encode_vectors(jt_, mp_)
        return
    bucket.zero()
    mean, logvar = model.encode_latent(jt_, mp_)
    (mean.sum() + logvar.sum()).backward()
batches = []
for i in range(6):
    trees = synthetic.make_batch(77 + i, B, vocab_size=385, mean_atoms=40.0)
    batches.append((T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])))
step(*batches[0]); step(*batches[1]); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
for b in batches[2:]: step(*b)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print("ms per fresh step (profiled): %.2f" % (1e3 * dt / 4))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
