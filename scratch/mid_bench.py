"""Mid-size plain products (rows of a tree level): time per launch, GPU kept warm by a long loop."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'jt-vae-tmcinvdes_amd'))
from fast_jtnn_hip import _capi as K
H, ld = 450, 452
def timeit(fn, n=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (600, 1000, 2000, 3000, 4000, 6000, 8000, 10000, 12000, 20000):
    A = torch.randn(M, ld, device='cuda'); W = torch.randn(H, ld, device='cuda'); C = torch.empty(M, ld, device='cuda')
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H))
    print('M=%6d: %6.1f us  %6.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
