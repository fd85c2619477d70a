"""Experiment: where the prefetched fresh-holder step loses time against device-resident holders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import parallel, synthetic, tensorize as T
from fast_jtnn_hip.prefetch import Prefetcher, warm_encoder_holders
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, "train")
model.overlap_streams = True
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
for i in range(12):
    trees = synthetic.make_batch(77 + i, B, vocab_size=385, mean_atoms=40.0)
    batches.append((T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])))
def unseen(b):
    return tuple(tuple(x.clone() if isinstance(x, torch.Tensor) else x for x in h) for h in b)
def loop(source, label):
    it = iter(source)
    step(*next(it)); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; wait = 0.0; issue = 0.0
    while True:
        a = time.perf_counter()
        try:
            jt_, mp_ = next(it)
        except StopIteration:
            break
        b = time.perf_counter()
        step(jt_, mp_)
        c = time.perf_counter()
        wait += b - a; issue += c - b; n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-34s %.2f ms/step  (waiting for a batch %.2f, issuing %.2f)" % (label, 1e3 * dt / n, 1e3 * wait / n, 1e3 * issue / n), flush=True)
# 1: same batch every step (device-resident holders)
loop([batches[0]] * 24, "resident, one batch")
# 2: 12 different batches, all prepared before the loop (twice through, second pass measured)
ws = [unseen(b) for b in batches]
for w in ws: warm_encoder_holders(model, w[0], w[1], train=mode == "train")
torch.cuda.synchronize()
loop(ws * 2, "resident, 12 different batches")
# 3: prefetched
profile = os.environ.get("PROFILE_STEPPING_THREAD") == "1"
for wk in ([workers] if profile else [workers, workers + 1, workers + 2, workers + 3]):
    pf = Prefetcher([unseen(b) for b in batches * 2], lambda b: warm_encoder_holders(model, b[0], b[1], train=mode == "train"), dev, workers=wk, depth=max(4, wk + 1))
    if profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    loop(pf, "prefetched, %d workers" % wk)
    pf.close()
    if profile:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# 4: prep alone, one thread
t0 = time.perf_counter()
for b in batches:
    w = unseen(b); warm_encoder_holders(model, w[0], w[1], train=mode == "train")
torch.cuda.synchronize()
print("prep alone: %.2f ms/batch" % (1e3 * (time.perf_counter() - t0) / len(batches)))
