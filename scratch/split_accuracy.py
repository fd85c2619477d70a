"""Accuracy of the split-bf16 large-product form of jtvae_gemm_nt against fp64, beside the fp32 form
(JTVAE_SPLIT_BF16=0): plain, bias + add + act, gathered A rows / scattered C rows."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
from fast_jtnn_hip import _capi as K
torch.manual_seed(0)
H, ld = 450, 452
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for name, mk in (("normal", lambda *s: torch.randn(*s, device="cuda")),
                 ("post-ReLU x small weights", None), ("wide dynamic range", None)):
    A = torch.zeros(M, ld, device="cuda"); W = torch.zeros(H, ld, device="cuda")
    if name == "normal":
        A[:, :H] = torch.randn(M, H, device="cuda"); W[:, :H] = torch.randn(H, H, device="cuda")
    elif name.startswith("post"):
        A[:, :H] = torch.relu(torch.randn(M, H, device="cuda")); W[:, :H] = 0.05 * torch.randn(H, H, device="cuda")
    else:
        A[:, :H] = torch.randn(M, H, device="cuda") * torch.exp(4 * torch.randn(M, H, device="cuda"))
        W[:, :H] = torch.randn(H, H, device="cuda") * torch.exp(4 * torch.randn(H, H, device="cuda"))
    Cb = torch.empty(M, ld, device="cuda")
    K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, Cb.data_ptr(), ld, M, H, H)
    torch.cuda.synchronize()
    ref = A[:, :H].double() @ W[:, :H].double().t()
    mag = A[:, :H].double().abs() @ W[:, :H].double().abs().t()
    err = (Cb[:, :H].double() - ref).abs()
    signed = ((Cb[:, :H].double() - ref) * ref.sign()).mean() / ref.abs().mean()  # > 0: magnitudes come out too large
    print("%-28s max |err| / sum|ab| %.3e   rms err / rms ref %.3e   mean signed err / mean |ref| %+.3e   pad cols zero: %s" % (
        name, float((err / mag).max()), float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(signed),
        bool((Cb[:, H:] == 0).all())))
# gathered / scattered rows with bias, addend and activation
A = torch.zeros(M, ld, device="cuda"); A[:, :H] = torch.randn(M, H, device="cuda")
W = torch.zeros(H, ld, device="cuda"); W[:, :H] = 0.05 * torch.randn(H, H, device="cuda")
rows = torch.randperm(M, device="cuda")[: M - 37].to(torch.int32).contiguous()
n = rows.numel()
bias = torch.randn(H, device="cuda"); add = torch.zeros(M, ld, device="cuda"); add[:, :H] = torch.randn(M, H, device="cuda")
Cb = torch.zeros(M, ld, device="cuda")
K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, Cb.data_ptr(), ld, n, H, H, bias=bias.data_ptr(), add=add.data_ptr(),
          ld_add=ld, act=K.ACT_TANH, a_rows=rows.data_ptr(), c_rows=rows.data_ptr())
torch.cuda.synchronize()
r = rows.long()
ref = torch.tanh(A[r, :H].double() @ W[:, :H].double().t() + bias.double() + add[r, :H].double())
err = (Cb[r, :H].double() - ref).abs().max()
print("row lists + bias + add + tanh    max |err| %.3e (values in [-1, 1])" % float(err))
# weight-gradient product: dW = X^T Y over M rows, bias column, row-pair lists
for M2 in (50000, 85181):
    X = torch.zeros(M2, ld, device="cuda"); X[:, :H] = torch.randn(M2, H, device="cuda") * (torch.rand(M2, H, device="cuda") < 0.5)
    Y = torch.zeros(M2, ld, device="cuda"); Y[:, :H] = torch.relu(torch.randn(M2, H, device="cuda"))
    dW = torch.empty(H, H, device="cuda"); db = torch.empty(H, device="cuda")
    K.gemm_tn(X.data_ptr(), ld, Y.data_ptr(), ld, dW.data_ptr(), H, M2, H, H, False, "cuda", dbias=db.data_ptr())
    torch.cuda.synchronize()
    ref = X[:, :H].double().t() @ Y[:, :H].double()
    mag = X[:, :H].double().abs().t() @ Y[:, :H].double().abs()
    err = (dW.double() - ref).abs()
    dbe = (db.double() - X[:, :H].double().sum(0)).abs().max() / X[:, :H].double().abs().sum(0).max()
    print("gemm_tn M=%d: max |err| / sum|xy| %.3e   rms err / rms ref %.3e   bias column %.3e" % (
        M2, float((err / mag).max()), float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(dbe)))
M2 = 30000
xr = torch.randint(0, M, (M2,), device="cuda", dtype=torch.int32); yr = torch.randint(0, M, (M2,), device="cuda", dtype=torch.int32)
dW = torch.empty(H, H, device="cuda")
K.gemm_tn(A.data_ptr(), ld, add.data_ptr(), ld, dW.data_ptr(), H, M2, H, H, False, "cuda", x_rows=xr.data_ptr(), y_rows=yr.data_ptr())
torch.cuda.synchronize()
ref = A[xr.long(), :H].double().t() @ add[yr.long(), :H].double()
print("gemm_tn row pairs: rms err / rms ref %.3e" % float((dW.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
