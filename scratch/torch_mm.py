import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
def bench(M, N, K, ld=None, reps=20):
    A = torch.randn(M, ld or K, device=dev)[:, :K]
    W = torch.randn(N, ld or K, device=dev)[:, :K]
    C = torch.empty(M, ld or N, device=dev)[:, :N]
    for _ in range(5): torch.mm(A, W.t(), out=C) if ld is None else torch.matmul(A, W.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if ld is None: torch.mm(A, W.t(), out=C)
        else: torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("M=%d N=%d K=%d ld=%s: %.1f us  %.1f TF" % (M, N, K, ld, us, 2.0 * M * N * K / us / 1e6), flush=True)
for M in (3608, 11572, 42943, 85181):
    bench(M, 450, 450)
    bench(M, 450, 450, ld=452)
bench(85181, 480, 480)
bench(8192, 4096, 4096)
# TN shape: dW[450,450] = X^T Y over M
for M in (42943, 85181):
    X = torch.randn(M, 450, device=dev); Y = torch.randn(M, 450, device=dev)
    for _ in range(5): torch.mm(X.t(), Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.mm(X.t(), Y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("TN M=%d: %.1f us %.1f TF" % (M, us, 2.0 * M * 450 * 450 / us / 1e6), flush=True)
