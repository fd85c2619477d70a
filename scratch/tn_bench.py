"""Micro-benchmark of jtvae_gemm_tn / jtvae_gemm_nt on the shapes the training step runs (GPU only).

  python3 scratch/tn_bench.py [tn|nt] [reps]
"""
import os, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
from fast_jtnn_hip import _capi as K

TN_SHAPES = [(385, 450, 450, 1), (1024, 450, 450, 1), (10310, 450, 450, 1), (17820, 450, 450, 1), (25000, 450, 450, 1), (40898, 450, 450, 1), (84000, 450, 450, 1), (84000, 450, 50, 1), (40000, 450, 485, 1),
             (13000, 450, 900, 1), (330000, 450, 450, 1), (1320000, 450, 450, 0), (5300000, 450, 450, 0)]
NT_SHAPES = [(25000, 450, 450), (84000, 450, 450), (84000, 450, 50), (330000, 450, 450), (1320000, 450, 450),
             (2000, 450, 450), (6000, 450, 450)]


def ld4(n):
    return (n + 3) // 4 * 4


def time_it(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "tn"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    if what == "tn":
        only = int(os.environ.get("TN_MIN_M", "0"))
        for M, N, Kk, bias in TN_SHAPES:
            if M < only:
                continue
            # rotate over enough operand copies that no call finds its rows in the 256 MB memory-side cache
            copies = max(1, min(8, int(700e6 // (4.0 * M * (ld4(N) + ld4(Kk)))) + 1))
            Xs = [torch.randn(M, ld4(N), generator=g).to(dev) for _ in range(copies)]
            Ys = [torch.randn(M, ld4(Kk), generator=g).to(dev) for _ in range(copies)]
            X, Y = Xs[0], Ys[0]
            dW = torch.zeros(N, ld4(Kk), device=dev)
            db = torch.zeros(N, device=dev) if bias else None
            nbytes = K.lib.jtvae_gemm_tn_workspace_bytes(M, N, Kk)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            turn = [0]

            def fn():
                i = turn[0] = (turn[0] + 1) % copies
                rc = K.lib.jtvae_gemm_tn(Xs[i].data_ptr(), ld4(N), None, Ys[i].data_ptr(), ld4(Kk), None,
                                         dW.data_ptr(), ld4(Kk), db.data_ptr() if bias else None, M, N, Kk, 0,
                                         ws.data_ptr(), nbytes, K.stream_ptr())
                assert rc == 0, rc

            ms = time_it(fn, reps)
            turn[0] = copies - 1
            fn()
            torch.cuda.synchronize()
            ref = X[:, :N].double().t() @ Y[:, :Kk].double()
            err = float((dW[:, :Kk].double() - ref).abs().max() / ref.abs().max())
            print("tn M=%-8d N=%-4d K=%-4d x%d %8.3f ms %7.1f TF/s  %6.1f GB/s  err %.1e" % (
                M, N, Kk, copies, ms, 2.0 * M * N * Kk / ms / 1e9, 4.0 * M * (N + Kk) / ms / 1e6, err), flush=True)
            del Xs, Ys
    else:
        for M, N, Kk in NT_SHAPES:
            A = torch.randn(M, ld4(Kk), generator=g).to(dev)
            W = torch.randn(N, ld4(Kk), generator=g).to(dev) * 0.05
            Cc = torch.zeros(M, ld4(N), device=dev)

            def fn():
                K.gemm_nt(A.data_ptr(), ld4(Kk), W.data_ptr(), ld4(Kk), Cc.data_ptr(), ld4(N), M, N, Kk)

            ms = time_it(fn, reps)
            print("nt M=%-8d N=%-4d K=%-4d %8.3f ms %7.1f TF/s  %6.1f GB/s" % (
                M, N, Kk, ms, 2.0 * M * N * Kk / ms / 1e9, 4.0 * M * (N + Kk) / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
