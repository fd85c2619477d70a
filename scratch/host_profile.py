"""scratch/host_profile.py [--batch B]: where the host time of one encoder step goes (cProfile over 200 resident
steps; the backward functions, which autograd runs on its own thread, are profiled there and timed one by one)."""
import cProfile
import os
import pstats
import sys
import time
from collections import defaultdict

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fresh-holders", "--no-kernel-timing", "--steps", "200", "--warmup", "5"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fast_jtnn_hip import jtnn_enc, jtnn_vae, mpn  # noqa: E402

spent = defaultdict(lambda: [0, 0.0])
bw_prof = cProfile.Profile()


def timed(cls):
    inner = cls.backward

    def backward(ctx, *grads):
        t0 = time.perf_counter()
        bw_prof.enable()
        try:
            return inner(ctx, *grads)
        finally:
            bw_prof.disable()
            s = spent[cls.__name__]
            s[0] += 1
            s[1] += time.perf_counter() - t0
    cls.backward = staticmethod(backward)


import torch  # noqa: E402
from fast_jtnn_hip import jtmpn, jtnn_dec  # noqa: E402

for mod in (jtnn_enc, mpn, jtnn_vae, jtmpn, jtnn_dec):
    for c in list(vars(mod).values()):
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c.__module__ == mod.__name__:
            timed(c)

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    for k, (n, t) in spent.items():
        print("%-28s %6d calls  %8.1f us each" % (k, n, 1e6 * t / max(n, 1)), file=sys.stderr)
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(30)
    st.sort_stats("cumulative").print_stats(40)
    sb = pstats.Stats(bw_prof, stream=sys.stderr)
    sb.sort_stats("tottime").print_stats(40)
    sb.sort_stats("cumulative").print_stats(40)
