import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'jt-vae-tmcinvdes_amd'))
import bench
from fast_jtnn_hip import parallel
model, jt, mp, stats = bench.build_problem(1234, 1024, 385, torch.device('cuda'))
bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])
s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
def latent(tv, mv):
    m = model
    mean = torch.cat([m.T_mean(tv), m.G_mean(mv)], dim=1)
    lv = torch.cat([-torch.abs(m.T_var(tv)), -torch.abs(m.G_var(mv))], dim=1)
    return mean, lv
def step_serial():
    bucket.zero()
    tv, _ = model.jtnn(*jt); mv = model.mpn(*mp)
    mean, lv = latent(tv, mv)
    (mean.sum() + lv.sum()).backward()
def step_overlap():
    bucket.zero()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        tv, _ = model.jtnn(*jt)
    with torch.cuda.stream(s2):
        mv = model.mpn(*mp)
    cur.wait_stream(s1); cur.wait_stream(s2)
    tv.record_stream(cur); mv.record_stream(cur)
    mean, lv = latent(tv, mv)
    (mean.sum() + lv.sum()).backward()
for name, fn in (("serial", step_serial), ("overlap", step_overlap), ("serial", step_serial), ("overlap", step_overlap)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s: %.3f ms/step  %.0f molecules/s" % (name, dt * 1e3, 1024 / dt), flush=True)
# gradients agree
step_serial(); torch.cuda.synchronize(); a = bucket.flat.clone()
step_overlap(); torch.cuda.synchronize(); b = bucket.flat.clone()
print("grad max abs diff", float((a - b).abs().max()), "equal", bool(torch.equal(a, b)))
