import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'jt-vae-tmcinvdes_amd'))
from fast_jtnn_hip import _capi as K
H, ld, M = 450, 452, 42943
A = torch.randn(M, ld, device='cuda'); W = torch.randn(H, ld, device='cuda'); C = torch.empty(M, ld, device='cuda')
add = torch.randn(M, ld, device='cuda')
for _ in range(5):
    K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H)
for _ in range(5):
    K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C.data_ptr(), ld, M, H, H, add=add.data_ptr(), ld_add=ld, act=K.ACT_SIGMOID)
torch.cuda.synchronize()
