import sys, numpy as np, torch
sys.path.insert(0, 'jt-vae-tmcinvdes_amd')
from fast_jtnn_hip import _capi as K
for (M, N, Kd) in ((4, 40, 96), (1, 450, 450), (130, 450, 450)):
    ld = K.round_ld(Kd)
    W = np.zeros((N, ld), np.float32)
    W[:, :Kd] = np.tile(np.arange(1, Kd + 1, dtype=np.float32), (N, 1)) + 1000 * np.arange(N, dtype=np.float32)[:, None]
    W[:, Kd:] = 7.5
    Wd = torch.from_numpy(W).cuda()
    bad = []
    for ks in range(Kd):
        A = np.full((M, ld), 0, np.float32); A[:, ks] = 1; A[:, Kd:] = 7.5
        Ad = torch.from_numpy(A).cuda()
        ldc = K.round_ld(N)
        C = torch.zeros((M, ldc), device='cuda')
        K.gemm_nt(Ad.data_ptr(), ld, Wd.data_ptr(), ld, C.data_ptr(), ldc, M, N, Kd)
        torch.cuda.synchronize()
        c = C.cpu().numpy()[:, :N]
        want = np.tile(W[:, ks], (M, 1))
        if not np.array_equal(c, want):
            bad.append(ks)
            if len(bad) <= 3:
                d = np.argwhere(c != want)
                print('shape', (M, N, Kd), 'k*', ks, 'n wrong', len(d), 'first', d[:3].tolist(), c[tuple(d[0])], want[tuple(d[0])])
    print((M, N, Kd), 'bad k:', bad[:20], len(bad))
# random full test with small ints (exact)
rs = np.random.RandomState(0)
M, N, Kd = 1, 450, 450
A = rs.randint(-3, 4, size=(M, Kd)).astype(np.float32); W = rs.randint(-3, 4, size=(N, Kd)).astype(np.float32)
ld = 452
Ad = torch.zeros((M, ld), device='cuda'); Ad[:, :Kd] = torch.from_numpy(A).cuda()
Wd = torch.zeros((N, ld), device='cuda'); Wd[:, :Kd] = torch.from_numpy(W).cuda()
C = torch.zeros((M, 452), device='cuda')
K.gemm_nt(Ad.data_ptr(), ld, Wd.data_ptr(), ld, C.data_ptr(), 452, M, N, Kd)
torch.cuda.synchronize()
ref = A @ W.T
d = C.cpu().numpy()[:, :N] - ref
print('int test max diff', np.abs(d).max(), 'n wrong', (d != 0).sum(), 'where', np.argwhere(d != 0)[:10].tolist())
