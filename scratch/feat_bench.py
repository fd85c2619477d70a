"""Feature-row products (K = 51 / 39) at the benchmark batch's row counts, warm."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'jt-vae-tmcinvdes_amd'))
from fast_jtnn_hip import _capi as K
H, ld = 450, 452
def timeit(fn, n=100, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, Kd in ((85181, 51), (40898, 39)):
    ldk = (Kd + 3) // 4 * 4
    A = (torch.rand(M, ldk, device='cuda') < 0.15).float(); W = torch.randn(H, ldk, device='cuda'); C = torch.empty(M, ld, device='cuda')
    b = torch.randn(H, device='cuda')
    ms = timeit(lambda: K.gemm_nt(A.data_ptr(), ldk, W.data_ptr(), ldk, C.data_ptr(), ld, M, H, Kd, bias=b.data_ptr()))
    print('gemm_nt M=%6d K=%2d: %6.1f us  out %.0f GB/s' % (M, Kd, ms * 1e3, M * ld * 4 / ms / 1e6))
    X = torch.randn(M, ld, device='cuda'); dW = torch.empty(H, ldk, device='cuda')
    ms = timeit(lambda: K.gemm_tn(X.data_ptr(), ld, A.data_ptr(), ldk, dW.data_ptr(), ldk, M, H, Kd, False, 'cuda'), n=50, warm=100)
    print('gemm_tn M=%6d K=%2d: %6.1f us  in  %.0f GB/s' % (M, Kd, ms * 1e3, M * ld * 4 / ms / 1e6))
