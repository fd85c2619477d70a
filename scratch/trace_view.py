import sys, numpy as np
f, us = sys.argv[1], float(sys.argv[2])
t = np.fromfile(f, dtype=np.uint64).reshape(-1, 4, 96)
nb = t.shape[0]
hw = t[:, :, 0]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
h = hw & np.uint64(0xFFFFFFFF)
simd = (h >> np.uint64(4)) & np.uint64(3); cu = (h >> np.uint64(8)) & np.uint64(15); sh = (h >> np.uint64(12)) & np.uint64(1); se = (h >> np.uint64(13)) & np.uint64(7)
# every XCD has its own counter base: rebase all stamps of a block on its XCD's first stamp
tt = t.astype(np.int64)
key = (xcc[:, 0].astype(int) * 1000 + se[:, 0].astype(int) * 100 + sh[:, 0].astype(int) * 16 + cu[:, 0].astype(int))
for x in np.unique(key):
    m = (key == x)
    base = tt[m][:, :, 1].min()
    tt[m, :, 1:] -= base
print("raw lifetime ticks of wave 0 of first blocks:", (tt[:5, 0, 80] - tt[:5, 0, 1]))
t = tt
st = t[:, :, 1]; en = t[:, :, 80]
t0 = st.min(); span = en.max() - t0
print("blocks", nb, "span ticks", span, "-> %.1f MHz if kernel=%.1f us" % (span / us, us))
tick = 0.01  # s_memtime counts a constant 100 MHz clock
life = (en - st) * tick
print("wave lifetime us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
# phases per K-step
nk = 15
comp = []; bar1 = []; stage = []; arrive = []; bar2 = []
for kt in range(nk):
    a = t[:, :, 2] if kt == 0 else t[:, :, 6 + 4 * (kt - 1)]
    b = t[:, :, 3 + 4 * kt]; c = t[:, :, 4 + 4 * kt]; d = t[:, :, 6 + 4 * kt]
    c2 = t[:, :, 5 + 4 * kt] if kt < nk - 1 else c
    comp.append((b - a) * tick); bar1.append((c - b) * tick); stage.append((d - c) * tick)
    arrive.append((c2 - c) * tick); bar2.append((d - c2) * tick)
comp = np.array(comp); bar1 = np.array(bar1); stage = np.array(stage); arrive = np.array(arrive); bar2 = np.array(bar2)
print("stage split: wait for tile + LDS store mean %.2f, second barrier mean %.2f" % (arrive[:14].mean(), bar2[:14].mean()))
for tq in range(3):
    print("epilogue t=%d: turn %.2f  store-issue %.2f" % (tq, ((t[:, :, 70 + 2 * tq] - (t[:, :, 6 + 4 * (nk - 1)] if tq == 0 else t[:, :, 69 + 2 * tq])) * tick).mean(), ((t[:, :, 71 + 2 * tq] - t[:, :, 70 + 2 * tq]) * tick).mean()))
pro = (t[:, :, 2] - st) * tick
epi = (en - t[:, :, 6 + 4 * (nk - 1)]) * tick
print("prologue us mean %.2f ; epilogue mean %.2f" % (pro.mean(), epi.mean()))
print("per K-step (tiles 0..13 full): compute mean %.2f us  wait-at-barrier mean %.2f  stage(next tile: vm wait + LDS store + barrier) mean %.2f" % (comp[:14].mean(), bar1[:14].mean(), stage[:14].mean()))
print("sum over K loop per wave: compute %.1f  barrier %.1f  stage %.1f" % (comp.sum(0).mean(), bar1.sum(0).mean(), stage.sum(0).mean()))
# start-time distribution: how many dispatch rounds
s_us = (st[:, 0] - t0) * tick
hist, edges = np.histogram(s_us, bins=12)
print("block start times histogram:", list(zip(edges[:-1].round(0), hist)))
# one CU: list its blocks in time order
key = (xcc[:, 0].astype(int) * 1000 + se[:, 0].astype(int) * 100 + sh[:, 0].astype(int) * 16 + cu[:, 0].astype(int))
ids, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(ids), "blocks per CU min/max", cnt.min(), cnt.max())
k0 = ids[len(ids) // 2]
sel = np.where(key == k0)[0]
sel = sel[np.argsort(st[sel, 0])]
print("CU key", k0, ": block, start us, end us, comp/bar/stage sums (wave0)")
for b in sel:
    print("  blk %5d  %7.1f -> %7.1f   comp %.1f bar %.1f stage %.1f  pro %.1f epi %.1f" % (b, (st[b, 0] - t0) * tick, (en[b, 0] - t0) * tick, comp[:, b, 0].sum(), bar1[:, b, 0].sum(), stage[:, b, 0].sum(), pro[b, 0], epi[b, 0]))
