"""Soak: many overlapped steps (two encoder streams + the helper stream of the tree encoder's backward pass) on
rotating batches, each compared bit for bit with the one-stream step on the same batch (GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import synthetic, tensorize as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, "train")
model.jtnn.crowded = True  # one tiling of the short products for both stream layouts: the comparison is bit for bit
model.overlap_min_messages = 0
batches = [(jt, mp)]
for i in range(3):
    trees = synthetic.make_batch(500 + i, B, vocab_size=385, mean_atoms=40.0)
    batches.append((T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])))


def step(b, overlap):
    model.overlap_streams = overlap
    for p in model.parameters():
        p.grad = None
    mean, logvar = model.encode_latent(*b)
    (mean.sum() + 0.5 * logvar.sum()).backward()
    return [mean.detach(), logvar.detach()] + [p.grad for p in model.parameters() if p.grad is not None]


refs = []
for b in batches:
    r = [t.clone() for t in step(b, False)]
    torch.cuda.synchronize()
    refs.append(r)
bad = 0
for it in range(iters):
    k = it % len(batches)
    got = step(batches[k], True)   # no host sync between steps: the next step's launches queue behind this one's
    ok = all(torch.equal(a, r) for a, r in zip(got, refs[k]))
    bad += 0 if ok else 1
torch.cuda.synchronize()
print("two-stream soak: %d steps on %d batches of %d, %d differ from the one-stream step" % (iters, len(batches), B, bad))
sys.exit(1 if bad else 0)
