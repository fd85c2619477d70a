import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'jt-vae-tmcinvdes_amd'))
from fast_jtnn_hip import _capi as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H, ld = 450, 452
for M in (42943, 85181):
    A = torch.randn(M, ld, device='cuda'); W = torch.randn(H, ld, device='cuda'); C = torch.empty(M, ld, device='cuda')
    add = torch.randn(M, ld, device='cuda')
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H))
    print('gemm_nt plain  M=%d: %.1f us  %.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H, add=add.data_ptr(), ld_add=ld, act=K.ACT_SIGMOID))
    print('gemm_nt add+sigmoid M=%d: %.1f us  %.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
M = 19 * 42943
X = torch.randn(M, ld, device='cuda'); Y = torch.randn(M, ld, device='cuda'); dW = torch.empty(H, H, device='cuda')
ms = timeit(lambda: K.gemm_tn(X.data_ptr(), ld, Y.data_ptr(), ld, dW.data_ptr(), H, M, H, H, False, 'cuda'), n=5)
print('gemm_tn M=%d: %.1f us  %.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
