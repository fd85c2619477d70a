"""Large products, GPU kept warm: time per launch of gemm_nt / gemm_tn at the row counts of the benchmark batches."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'jt-vae-tmcinvdes_amd'))
from fast_jtnn_hip import _capi as K
H, ld = 450, 452
def timeit(fn, n=60, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (40898, 85181, 340000):
    A = torch.randn(M, ld, device='cuda'); W = torch.randn(H, ld, device='cuda'); C = torch.empty(M, ld, device='cuda')
    add = torch.randn(M, ld, device='cuda')
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H))
    print('gemm_nt plain     M=%7d: %7.1f us  %6.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H, add=add.data_ptr(), ld_add=ld, act=K.ACT_RELU))
    print('gemm_nt add+relu  M=%7d: %7.1f us  %6.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
    dW = torch.empty(H, H, device='cuda')
    ms = timeit(lambda: K.gemm_tn(A.data_ptr(), ld, add.data_ptr(), ld, dW.data_ptr(), H, M, H, H, False, 'cuda'), n=30, warm=100)
    print('gemm_tn           M=%7d: %7.1f us  %6.1f TFLOP/s' % (M, ms * 1e3, 2 * M * H * H / ms / 1e9))
