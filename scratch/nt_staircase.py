"""Experiment: duration of jtvae_gemm_nt against the row count, to see how much the last partial round of
workgroups costs (GPU only)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "jt-vae-tmcinvdes_amd"))
import torch
from fast_jtnn_hip import _capi as K
from tn_bench import time_it, ld4

dev = torch.device("cuda:0")
N = Kk = 450
W = torch.randn(N, ld4(Kk), device=dev) * 0.05
for M in (26112, 39168, 52224, 65536, 78592, 80000, 82000, 85181, 90000, 98304, 104704, 104960, 111000, 131072):
    A = torch.randn(M, ld4(Kk), device=dev)
    Cc = torch.zeros(M, ld4(N), device=dev)
    ms = time_it(lambda: K.gemm_nt(A.data_ptr(), ld4(Kk), W.data_ptr(), ld4(Kk), Cc.data_ptr(), ld4(N), M, N, Kk), 30)
    tiles = (M + 127) // 128 * 5
    print("M=%-7d tiles %5d rounds %.2f  %.1f us  %.1f TF/s" % (M, tiles, tiles / 1024.0, ms * 1e3, 2.0 * M * N * Kk / ms / 1e9), flush=True)
