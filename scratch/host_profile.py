"""Experiment: where the host time of one full training step goes (cProfile over a few steps)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
import bench
from fast_jtnn_hip import trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else "train_full"
dev = torch.device("cuda", 0)
model, jt, mp, xb, stats = bench.build_problem(1234, B, 385, dev, mode)
tr = trainer.Trainer(model, trainer.Schedule(beta=0.005, print_iter=10**9, save_iter=10**9, anneal_iter=10**9, kl_anneal_iter=10**9), log=lambda *a: None)
for _ in range(2): tr.step(xb)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.step(xb)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
