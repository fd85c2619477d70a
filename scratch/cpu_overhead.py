import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'jt-vae-tmcinvdes_amd'))
import bench
from fast_jtnn_hip import parallel
model, jt, mp, stats = bench.build_problem(1234, 1024, 385, torch.device('cuda'))
bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])
def step():
    bucket.zero()
    mean, logvar = model.encode_latent(jt, mp)
    (mean.sum() + logvar.sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
# CPU issue time: run 5 steps without sync, measure host time until all issued, then sync time
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per step %.2f ms ; total per step %.2f ms' % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
