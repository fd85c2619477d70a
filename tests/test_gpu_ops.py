"""GPU: every kernel entry point of the C ABI against a CPU fp64/fp32 restatement of the op."""
import numpy as np
import pytest
import torch

from common import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from fast_jtnn_hip import _capi
    assert torch.cuda.is_available()
    return _capi


# Raw pointers cross the C ABI, so every device tensor must outlive the launch that reads it:
# a temporary freed right after .data_ptr() can be handed to the next allocation and overwritten
# before the kernel runs.  Helpers park their tensors here until the test ends.
_KEEP = []


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def keep(t):
    _KEEP.append(t)
    return t


def dev_rows(x, ld):
    """numpy [n, H] -> padded device buffer [n, ld] with NaN-free garbage in the padding."""
    n, h = x.shape
    buf = torch.full((n, ld), 7.5, dtype=torch.float32, device="cuda")
    buf[:, :h] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    return keep(buf)


def dev(x):
    return keep(torch.from_numpy(np.ascontiguousarray(x)).cuda())


@pytest.mark.parametrize("M,N,Kd", [(1, 450, 450), (129, 450, 450), (1000, 450, 450), (257, 96, 96),
                                    (300, 450, 51), (64, 450, 39), (515, 33, 7), (10111, 450, 450), (12000, 56, 450)])
def test_gemm_nt_plain(K, M, N, Kd):
    rs = np.random.RandomState(M + N + Kd)
    A = rs.standard_normal((M, Kd)).astype(np.float32)
    W = rs.standard_normal((N, Kd)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    add = rs.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias + add
    lda = K.round_ld(Kd)
    # (operands aligned?, ldc): float4 loads + float4 epilogue, float4 loads + scalar epilogue, all scalar
    for aligned, ldc in ((True, K.round_ld(N)), (True, K.round_ld(N) + 1), (False, K.round_ld(N))):
        if aligned:
            Ad, Wd = dev_rows(A, lda), dev_rows(W, lda)
            a_ld = w_ld = lda
        else:
            Ad, Wd = dev(A), dev(W)
            a_ld = w_ld = Kd
        C = torch.full((M, ldc), -3.0, device="cuda")
        pre = torch.full((M, ldc), -3.0, device="cuda")
        K.gemm_nt(Ad.data_ptr(), a_ld, Wd.data_ptr(), w_ld, C.data_ptr(), ldc, M, N, Kd,
                  bias=dev(bias).data_ptr(), add=dev_rows(add, ldc).data_ptr(), ld_add=ldc,
                  act=K.ACT_RELU, pre_out=pre.data_ptr(), ld_pre=ldc)
        torch.cuda.synchronize()
        assert rel_err(pre[:, :N].cpu().numpy(), ref) < 2e-6
        assert rel_err(C[:, :N].cpu().numpy(), np.maximum(ref, 0)) < 2e-6
        if N < ldc == K.round_ld(N):  # the padding of the canonical layout is written as zeros
            assert float(C[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,pad", [(333, 0), (777, 0), (777, 3), (10300, 0)])
def test_gemm_nt_table_addend_and_activations(K, M, pad):
    rs = np.random.RandomState(5)
    N, Kd, V = 450, 450, 17
    ld = K.round_ld(N) + pad  # pad 3: rows not 16-byte aligned, scalar kernels
    A = rs.standard_normal((M, Kd)).astype(np.float32) * 0.1
    W = rs.standard_normal((N, Kd)).astype(np.float32) * 0.1
    tab = rs.standard_normal((V, N)).astype(np.float32)
    idx = rs.randint(0, V, size=M).astype(np.int32)
    pre = A.astype(np.float64) @ W.astype(np.float64).T + tab[idx]
    for act, fn in ((K.ACT_SIGMOID, lambda v: 1 / (1 + np.exp(-v))), (K.ACT_TANH, np.tanh), (K.ACT_NONE, lambda v: v)):
        C = torch.empty((M, ld), device="cuda")
        K.gemm_nt(dev_rows(A, ld).data_ptr(), ld, dev_rows(W, ld).data_ptr(), ld, C.data_ptr(), ld, M, N, Kd,
                  add=dev_rows(tab, ld).data_ptr(), ld_add=ld, add_idx=dev(idx).data_ptr(),
                  act=act)
        torch.cuda.synchronize()
        assert rel_err(C[:, :N].cpu().numpy(), fn(pre)) < 3e-6


@pytest.mark.parametrize("M,pad", [(200, 0), (900, 0), (900, 1), (10300, 0)])
def test_gemm_nt_gru_blend_and_mul_acc(K, M, pad):
    rs = np.random.RandomState(6)
    N = 450
    ld = K.round_ld(N) + pad  # pad 1: scalar kernels
    G = rs.standard_normal((M, N)).astype(np.float32) * 0.1
    W = rs.standard_normal((N, N)).astype(np.float32) * 0.1
    tab = rs.standard_normal((9, N)).astype(np.float32)
    idx = rs.randint(0, 9, size=M).astype(np.int32)
    z = rs.rand(M, N).astype(np.float32)
    s = rs.standard_normal((M, N)).astype(np.float32)
    p_ref = np.tanh(G.astype(np.float64) @ W.astype(np.float64).T + tab[idx])
    h_ref = (1 - z) * s + z * p_ref
    h_ref[0] = 0
    P = torch.empty((M, ld), device="cuda")
    Hn = torch.empty((M, ld), device="cuda")
    K.gemm_nt(dev_rows(G, ld).data_ptr(), ld, dev_rows(W, ld).data_ptr(), ld, P.data_ptr(), ld, M, N, N,
              add=dev_rows(tab, ld).data_ptr(), ld_add=ld, add_idx=dev(idx).data_ptr(),
              mode=K.EPI_GRU_BLEND, aux0=dev_rows(z, ld).data_ptr(), aux1=dev_rows(s, ld).data_ptr(),
              out2=Hn.data_ptr(), ld_aux=ld, zero_row=0)
    torch.cuda.synchronize()
    assert rel_err(P[:, :N].cpu().numpy(), p_ref) < 3e-6
    assert rel_err(Hn[:, :N].cpu().numpy(), h_ref) < 3e-6
    assert float(Hn[:, N:].abs().max()) == 0.0 and float(Hn[0].abs().max()) == 0.0
    # multiply-accumulate epilogue
    t = rs.standard_normal((M, N)).astype(np.float32)
    acc0 = rs.standard_normal((M, N)).astype(np.float32)
    acc = dev_rows(acc0, ld)
    C = torch.empty((M, ld), device="cuda")
    K.gemm_nt(dev_rows(G, ld).data_ptr(), ld, dev_rows(W, ld).data_ptr(), ld, C.data_ptr(), ld, M, N, N,
              mode=K.EPI_MUL_ACC, aux0=dev_rows(t, ld).data_ptr(), out2=acc.data_ptr(), ld_aux=ld)
    torch.cuda.synchronize()
    c_ref = G.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(C[:, :N].cpu().numpy(), c_ref) < 3e-6
    assert rel_err(acc[:, :N].cpu().numpy(), acc0 + c_ref * t) < 3e-6


@pytest.mark.parametrize("M,N,Kd", [(1000, 450, 450), (70000, 450, 450), (5000, 450, 39), (31, 96, 51),
                                    (4097, 450, 44), (700, 130, 96), (300, 450, 480)])
def test_gemm_tn_and_bias_column(K, M, N, Kd):
    rs = np.random.RandomState(M)
    X = rs.standard_normal((M, N)).astype(np.float32)
    Y = rs.standard_normal((M, Kd)).astype(np.float32)
    ref = X.astype(np.float64).T @ Y.astype(np.float64)
    ldx, ldy = K.round_ld(N), K.round_ld(Kd)
    base = rs.standard_normal((N, Kd + 5)).astype(np.float32)
    dW = torch.from_numpy(base.copy()).cuda()
    Xd, Yd = dev_rows(X, ldx), dev_rows(Y, ldy)
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW.data_ptr(), Kd + 5, M, N, Kd, False, "cuda")
    torch.cuda.synchronize()
    got = dW.cpu().numpy()
    scale = np.sqrt(M)
    assert np.abs(got[:, :Kd] - ref).max() / scale < 2e-5
    assert np.array_equal(got[:, Kd:], base[:, Kd:])  # columns beyond K untouched (strided destination)
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW.data_ptr(), Kd + 5, M, N, Kd, True, "cuda")
    torch.cuda.synchronize()
    assert np.abs(dW.cpu().numpy()[:, :Kd] - 2 * ref).max() / scale < 4e-5
    # bitwise reproducible
    dW2 = torch.zeros_like(dW)
    dW3 = torch.zeros_like(dW)
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW2.data_ptr(), Kd + 5, M, N, Kd, False, "cuda")
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW3.data_ptr(), Kd + 5, M, N, Kd, False, "cuda")
    torch.cuda.synchronize()
    assert torch.equal(dW2, dW3)
    # the same column sums ride along with the contraction (bias gradient), without disturbing dW
    db = torch.full((N,), 3.0, device="cuda")
    dW4 = torch.from_numpy(base.copy()).cuda()
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW4.data_ptr(), Kd + 5, M, N, Kd, False, "cuda",
              dbias=db.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(db.cpu().numpy() - X.astype(np.float64).sum(0)).max() / scale < 2e-5
    assert torch.equal(dW4[:, :Kd], dW2[:, :Kd]) and np.array_equal(dW4.cpu().numpy()[:, Kd:], base[:, Kd:])
    K.gemm_tn(Xd.data_ptr(), ldx, Yd.data_ptr(), ldy, dW4.data_ptr(), Kd + 5, M, N, Kd, True, "cuda",
              dbias=db.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(db.cpu().numpy() - 2 * X.astype(np.float64).sum(0)).max() / scale < 4e-5


def test_copy_rows(K):
    rs = np.random.RandomState(1)
    W = rs.standard_normal((450, 489)).astype(np.float32)
    Wd = torch.from_numpy(W).cuda()
    P = torch.empty((450, 40), device="cuda")
    K.copy_rows(Wd.data_ptr(), 489, P.data_ptr(), 40, 450, 39)
    torch.cuda.synchronize()
    assert np.array_equal(P[:, :39].cpu().numpy(), W[:, :39]) and float(P[:, 39].abs().max()) == 0.0


def test_byte_feature_rows_widen_to_the_floats_they_came_from(K):
    """One-hot feature rows shipped as bytes: the padded device rows carry the bits of the float upload; rows
    that are not whole numbers in [0, 255] take the float route."""
    from fast_jtnn_hip import graphprep as G

    rs = np.random.RandomState(3)
    x = (rs.rand(1237, 39) < 0.1).astype(np.float32)
    x[5, 7] = 255.0
    rows, packed = G._upload_features(x, "cuda")
    assert rows is None and packed.dtype == torch.uint8 and np.array_equal(packed.cpu().numpy(), x.astype(np.uint8))
    got = G._widen_features(packed)
    torch.cuda.synchronize()
    assert got.shape == (1237, 40) and np.array_equal(got[:, :39].cpu().numpy(), x)
    assert float(got[:, 39].abs().max()) == 0.0
    stage = np.empty(x.shape, dtype=np.uint8)
    assert K.prep_features_to_u8(x, stage) and np.array_equal(stage, x.astype(np.uint8))
    y = x.copy()
    y[1000, 38] = 0.5
    assert not K.prep_features_to_u8(y, stage)
    got, packed = G._upload_features(y, "cuda")
    assert packed is None
    assert np.array_equal(got[:, :39].cpu().numpy(), y) and float(got[:, 39].abs().max()) == 0.0
    assert K.lib.jtvae_widen_u8_rows(None, got.data_ptr(), 40, 5, 39, None) == K.E_NULL
    assert K.lib.jtvae_widen_u8_rows(got.data_ptr(), got.data_ptr(), 36, 5, 39, None) == K.E_SHAPE
    assert K.lib.jtvae_widen_u8_rows(None, None, 40, 0, 39, None) == 0


def test_pack_batch(K):
    """Several weight blocks copied / transposed into padded row buffers by one launch."""
    rs = np.random.RandomState(2)
    W1 = rs.standard_normal((450, 489)).astype(np.float32)
    W2 = rs.standard_normal((450, 900)).astype(np.float32)
    W3 = rs.standard_normal((56, 450)).astype(np.float32)
    d1, d2, d3 = dev(W1), dev(W2), dev(W3)
    specs = [(d1, W1, 0, 39, 0), (d1, W1, 39, 450, 0), (d1, W1, 39, 450, 1), (d2, W2, 450, 450, 1),
             (d2, W2, 0, 450, 0), (d3, W3, 0, 450, 1), (d3, W3, 0, 450, 0)] * 3  # 21 items: two launches
    items, outs = [], []
    for d, W, c0, nc, tr in specs:
        rows, ld = (nc, K.round_ld(W.shape[0])) if tr else (W.shape[0], K.round_ld(nc))
        buf = keep(torch.full((rows, ld), 9.0, device="cuda"))
        outs.append(buf)
        items.append((K.ptr(d, c0), W.shape[1], buf.data_ptr(), ld, W.shape[0], nc, tr))
    K.pack_batch(items)
    torch.cuda.synchronize()
    for (d, W, c0, nc, tr), buf in zip(specs, outs):
        blk = W[:, c0:c0 + nc]
        ref = blk.T if tr else blk
        got = buf.cpu().numpy()
        assert np.array_equal(got[:, :ref.shape[1]], ref)
        assert not got[:, ref.shape[1]:].any()


def random_csr(rs, n_rows, n_src, max_deg):
    deg = rs.randint(0, max_deg + 1, size=n_rows)
    rp = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    col = rs.randint(1, n_src, size=int(rp[-1]))
    return rp, col


@pytest.mark.parametrize("H", [450, 96, 7])
def test_csr_gather_sum_variants(K, H):
    rs = np.random.RandomState(H)
    ld = K.round_ld(H)
    n0, n1, n_rows = 40, 70, 333
    s0 = rs.standard_normal((n0, H)).astype(np.float32)
    s1 = rs.standard_normal((n1, H)).astype(np.float32)
    cat = np.concatenate([s0, s1])
    rp, col = random_csr(rs, n_rows, n0 + n1, 9)
    ref = np.zeros((n_rows, H), dtype=np.float64)
    for i in range(n_rows):
        for e in range(rp[i], rp[i + 1]):
            ref[i] += cat[col[e]]
    rpd = dev(rp.astype(np.int32))
    cold = dev(col.astype(np.int32))
    out = torch.full((n_rows, ld), 9.0, device="cuda")
    K.csr_gather_sum(out.data_ptr(), dev_rows(s0, ld).data_ptr(), n0, dev_rows(s1, ld).data_ptr(), rpd.data_ptr(),
                     cold.data_ptr(), n_rows, H, ld)
    torch.cuda.synchronize()
    assert rel_err(out[:, :H].cpu().numpy(), ref) < 1e-6
    if ld > H:
        assert float(out[:, H:].abs().max()) == 0.0
    # single source + relu mask + accumulate + second accumulator
    mask = rs.standard_normal((n_rows, H)).astype(np.float32)
    prev = rs.standard_normal((n_rows, H)).astype(np.float32)
    acc2 = rs.standard_normal((n_rows, H)).astype(np.float32)
    outd, acc2d = dev_rows(prev, ld), dev_rows(acc2, ld)
    K.csr_gather_sum(outd.data_ptr(), dev_rows(cat, ld).data_ptr(), n0 + n1, None, rpd.data_ptr(), cold.data_ptr(),
                     n_rows, H, ld, relu_mask=dev_rows(mask, ld).data_ptr(), accumulate=True,
                     acc2=acc2d.data_ptr())
    torch.cuda.synchronize()
    want = ref * (mask > 0) + prev
    assert rel_err(outd[:, :H].cpu().numpy(), want) < 1e-6
    assert rel_err(acc2d[:, :H].cpu().numpy(), acc2 + want) < 1e-6
    # ReLU taken on read: of the second source only when there are two, of the only source otherwise
    for two in (True, False):
        act = np.concatenate([s0, np.maximum(s1, 0)]) if two else np.maximum(cat, 0)
        ref_r = np.zeros((n_rows, H), dtype=np.float64)
        for i in range(n_rows):
            for e in range(rp[i], rp[i + 1]):
                ref_r[i] += act[col[e]]
        o = torch.empty((n_rows, ld), device="cuda")
        if two:
            K.csr_gather_sum(o.data_ptr(), dev_rows(s0, ld).data_ptr(), n0, dev_rows(s1, ld).data_ptr(), rpd.data_ptr(),
                             cold.data_ptr(), n_rows, H, ld, relu_src=True)
        else:
            K.csr_gather_sum(o.data_ptr(), dev_rows(cat, ld).data_ptr(), n0 + n1, None, rpd.data_ptr(), cold.data_ptr(),
                             n_rows, H, ld, relu_src=True)
        torch.cuda.synchronize()
        assert rel_err(o[:, :H].cpu().numpy(), ref_r) < 1e-6


def test_segment_mean_and_backward(K):
    rs = np.random.RandomState(3)
    H, ld = 450, 452
    lens = rs.randint(1, 60, size=50)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    n = int(lens.sum())
    rows = rs.standard_normal((n, H)).astype(np.float32)
    seg = np.stack([starts, lens], 1).astype(np.int32)
    segd = dev(seg)
    out = torch.empty((50, ld), device="cuda")
    K.segment_mean(out.data_ptr(), dev_rows(rows, ld).data_ptr(), segd.data_ptr(), 50, H, ld)
    ref = np.stack([rows[s:s + l].astype(np.float64).mean(0) for s, l in zip(starts, lens)])
    torch.cuda.synchronize()
    assert rel_err(out[:, :H].cpu().numpy(), ref) < 1e-6
    dout = rs.standard_normal((50, H)).astype(np.float32)
    seg_of = np.repeat(np.arange(50), lens).astype(np.int32)
    seg_of[5] = -1
    drows = torch.empty((n, ld), device="cuda")
    K.segment_mean_bwd(drows.data_ptr(), dev_rows(dout, ld).data_ptr(), dev_rows(rows, ld).data_ptr(),
                       segd.data_ptr(), dev(seg_of).data_ptr(), n, H, ld)
    torch.cuda.synchronize()
    want = dout[np.maximum(seg_of, 0)] / lens[np.maximum(seg_of, 0)][:, None] * (rows > 0)
    want[5] = 0
    assert rel_err(drows[:, :H].cpu().numpy(), want) < 1e-6


def test_gru_kernels(K):
    rs = np.random.RandomState(4)
    H, ld, NM, V = 450, 452, 211, 13
    h = rs.standard_normal((NM, H)).astype(np.float32) * 0.5
    h[0] = 0
    Uh = rs.standard_normal((NM, H)).astype(np.float32)
    Pr = rs.standard_normal((V, H)).astype(np.float32)
    Pz = rs.standard_normal((V, H)).astype(np.float32)
    Ph = rs.standard_normal((V, H)).astype(np.float32)
    wid = rs.randint(0, V, size=NM).astype(np.int32)
    rp, col = random_csr(rs, NM, NM, 5)
    rp[1:] -= rp[1]  # row 0 reads nothing
    col = col[-int(rp[-1]):] if rp[-1] else col[:0]
    sig = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    s_ref = np.zeros((NM, H)); g_ref = np.zeros((NM, H)); t_ref = np.zeros((NM, H))
    for m in range(NM):
        for e in range(rp[m], rp[m + 1]):
            j = col[e]
            r = sig(Pr[wid[m]].astype(np.float64) + Uh[j])
            s_ref[m] += h[j]; g_ref[m] += r * h[j]; t_ref[m] += r * (1 - r) * h[j]
    dv = lambda x: dev_rows(x.astype(np.float32), ld)  # noqa: E731
    i32 = lambda x: dev(np.asarray(x, dtype=np.int32))  # noqa: E731
    hd, Ud, Prd, widd, rpd, cold = dv(h), dv(Uh), dv(Pr), i32(wid), i32(rp), i32(col)
    s, g, t = (torch.empty((NM, ld), device="cuda") for _ in range(3))
    K.gru_gate_gather(hd.data_ptr(), Ud.data_ptr(), 0, 0, None, Prd.data_ptr(), widd.data_ptr(), rpd.data_ptr(),
                      cold.data_ptr(), None, NM, s.data_ptr(), g.data_ptr(), t.data_ptr(), H, ld)
    s2, g2 = (torch.empty((NM, ld), device="cuda") for _ in range(2))
    K.gru_gate_gather(hd.data_ptr(), Ud.data_ptr(), 0, 0, None, Prd.data_ptr(), widd.data_ptr(), rpd.data_ptr(),
                      cold.data_ptr(), None, NM, s2.data_ptr(), g2.data_ptr(), None, H, ld)
    torch.cuda.synchronize()
    assert rel_err(s[:, :H].cpu().numpy(), s_ref) < 1e-6
    assert rel_err(g[:, :H].cpu().numpy(), g_ref) < 2e-6
    assert rel_err(t[:, :H].cpu().numpy(), t_ref) < 2e-6
    assert torch.equal(s, s2) and torch.equal(g, g2)
    # iteration 0
    z, p, h1 = (torch.empty((NM, ld), device="cuda") for _ in range(3))
    K.gru_iter0(dv(Pz).data_ptr(), dv(Ph).data_ptr(), widd.data_ptr(), z.data_ptr(), p.data_ptr(), h1.data_ptr(),
                NM, H, ld, 0)
    torch.cuda.synchronize()
    z_ref, p_ref = sig(Pz[wid].astype(np.float64)), np.tanh(Ph[wid].astype(np.float64))
    h_ref = z_ref * p_ref
    h_ref[0] = 0
    assert rel_err(z[:, :H].cpu().numpy(), z_ref) < 2e-6 and rel_err(p[:, :H].cpu().numpy(), p_ref) < 2e-6
    assert rel_err(h1[:, :H].cpu().numpy(), h_ref) < 2e-6
    # pointwise backward
    dh = rs.standard_normal((NM, H)).astype(np.float32)
    zz, pp, ss = rs.rand(NM, H).astype(np.float32), np.tanh(rs.standard_normal((NM, H))).astype(np.float32), s_ref
    dz, dp, ds = (torch.empty((NM, ld), device="cuda") for _ in range(3))
    K.gru_bwd_pointwise(dv(dh).data_ptr(), dv(zz).data_ptr(), dv(pp).data_ptr(), dv(ss).data_ptr(), 0, 1, None, None,
                        NM, dz.data_ptr(), dp.data_ptr(), ds.data_ptr(), H, ld, 0)
    torch.cuda.synchronize()
    d = dh.astype(np.float64).copy(); d[0] = 0
    assert rel_err(dz[:, :H].cpu().numpy(), d * (pp - ss) * zz * (1 - zz)) < 2e-6
    assert rel_err(dp[:, :H].cpu().numpy(), d * zz * (1 - pp.astype(np.float64) ** 2)) < 2e-6
    assert rel_err(ds[:, :H].cpu().numpy(), d * (1 - zz)) < 2e-6
    # gather backward on the transposed graph
    from fast_jtnn_hip import graphprep
    trp, tcol = graphprep.transpose_csr(rp.astype(np.int64), col.astype(np.int64), NM)
    dg = rs.standard_normal((NM, H)).astype(np.float32)
    dsv = rs.standard_normal((NM, H)).astype(np.float32)
    du_ref = np.zeros((NM, H)); dh_ref = np.zeros((NM, H))
    for m in range(NM):
        for e in range(rp[m], rp[m + 1]):
            j = col[e]
            r = sig(Pr[wid[m]].astype(np.float64) + Uh[j])
            du_ref[j] += dg[m] * r * (1 - r) * h[j]
            dh_ref[j] += dsv[m] + dg[m] * r
    du, dhd = (torch.empty((NM, ld), device="cuda") for _ in range(2))
    K.gru_bwd_gather(hd.data_ptr(), Ud.data_ptr(), 0, 0, None, None, Prd.data_ptr(), widd.data_ptr(),
                     dv(dg).data_ptr(), dv(dsv).data_ptr(), i32(trp).data_ptr(), i32(tcol).data_ptr(), None, NM,
                     du.data_ptr(), dhd.data_ptr(), H, ld)
    torch.cuda.synchronize()
    assert rel_err(du[:, :H].cpu().numpy(), du_ref) < 3e-6
    assert rel_err(dhd[:, :H].cpu().numpy(), dh_ref) < 3e-6


def test_row_dot(K):
    rs = np.random.RandomState(8)
    H, ld, NC, B = 450, 452, 77, 9
    q = rs.standard_normal((B, H)).astype(np.float32)
    c = rs.standard_normal((NC, H)).astype(np.float32)
    own = rs.randint(0, B, size=NC).astype(np.int32)
    sc = torch.empty(NC, device="cuda")
    K.row_dot(sc.data_ptr(), dev_rows(q, ld).data_ptr(), dev(own).data_ptr(),
              dev_rows(c, ld).data_ptr(), NC, H, ld)
    torch.cuda.synchronize()
    assert rel_err(sc.cpu().numpy(), (q[own].astype(np.float64) * c).sum(1)) < 2e-6


def test_row_list_variants(K):
    """Gathered A rows / scattered C rows in gemm_nt, row-pair lists in gemm_tn, active-row lists and
    per-iteration slabs in the GRU kernels."""
    rs = np.random.RandomState(11)
    H, ld, NM, n = 450, 452, 300, 77
    rows = np.sort(rs.choice(NM, size=n, replace=False)).astype(np.int32)
    A = rs.standard_normal((NM, H)).astype(np.float32) * 0.2
    W = rs.standard_normal((H, H)).astype(np.float32) * 0.2
    base = rs.standard_normal((NM, H)).astype(np.float32)
    C = dev_rows(base, ld)
    rd = dev(rows)
    K.gemm_nt(dev_rows(A, ld).data_ptr(), ld, dev_rows(W, ld).data_ptr(), ld, C.data_ptr(), ld, n, H, H,
              a_rows=rd.data_ptr(), c_rows=rd.data_ptr(), add=C.data_ptr(), ld_add=ld)
    torch.cuda.synchronize()
    want = base.astype(np.float64).copy()
    want[rows] += A[rows].astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(C[:, :H].cpu().numpy(), want) < 3e-6  # untouched rows keep their value exactly
    # compact A, scattered C
    Cc = dev_rows(base, ld)
    K.gemm_nt(dev_rows(A[rows], ld).data_ptr(), ld, dev_rows(W, ld).data_ptr(), ld, Cc.data_ptr(), ld, n, H, H,
              c_rows=rd.data_ptr())
    torch.cuda.synchronize()
    want2 = base.astype(np.float64).copy()
    want2[rows] = A[rows].astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(Cc[:, :H].cpu().numpy(), want2) < 3e-6
    # gemm_tn over row pairs
    M = 1000
    xr = rs.randint(0, NM, size=M).astype(np.int32)
    yr = rs.randint(0, NM, size=M).astype(np.int32)
    Y = rs.standard_normal((NM, H)).astype(np.float32)
    dW = torch.zeros((H, H), device="cuda")
    K.gemm_tn(dev_rows(A, ld).data_ptr(), ld, dev_rows(Y, ld).data_ptr(), ld, dW.data_ptr(), H, M, H, H, False, "cuda",
              x_rows=dev(xr).data_ptr(), y_rows=dev(yr).data_ptr())
    torch.cuda.synchronize()
    ref = A[xr].astype(np.float64).T @ Y[yr].astype(np.float64)
    assert rel_err(dW.cpu().numpy(), ref) < 3e-6
    # gate gather on a row list with slabs: neighbour j is read from slab min(t, sf[j])
    T_, V = 4, 9
    hs = rs.standard_normal((T_, NM, H)).astype(np.float32) * 0.5
    us = rs.standard_normal((T_, NM, H)).astype(np.float32)
    sf = rs.randint(1, T_ + 2, size=NM).astype(np.int32)
    Pr = rs.standard_normal((V, H)).astype(np.float32)
    wid = rs.randint(0, V, size=NM).astype(np.int32)
    rp, col = random_csr(rs, NM, NM, 4)
    t_iter = 2
    hsd = keep(torch.zeros((T_, NM, ld), device="cuda")); hsd[:, :, :H] = torch.from_numpy(hs).cuda()
    usd = keep(torch.zeros((T_, NM, ld), device="cuda")); usd[:, :, :H] = torch.from_numpy(us).cuda()
    s0 = rs.standard_normal((NM, H)).astype(np.float32)
    sd, gd, td = dev_rows(s0, ld), dev_rows(s0, ld), dev_rows(s0, ld)
    K.gru_gate_gather(hsd.data_ptr(), usd.data_ptr(), NM * ld, t_iter, dev(sf).data_ptr(), dev_rows(Pr, ld).data_ptr(),
                      dev(wid).data_ptr(), dev(rp.astype(np.int32)).data_ptr(), dev(col.astype(np.int32)).data_ptr(),
                      rd.data_ptr(), n, sd.data_ptr(), gd.data_ptr(), td.data_ptr(), H, ld)
    torch.cuda.synchronize()
    sig = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    s_ref, g_ref = s0.astype(np.float64).copy(), s0.astype(np.float64).copy()
    for m in rows:
        s_ref[m] = 0; g_ref[m] = 0
        for e in range(rp[m], rp[m + 1]):
            j = col[e]; tau = min(t_iter, sf[j])
            r = sig(Pr[wid[m]].astype(np.float64) + us[tau, j])
            s_ref[m] += hs[tau, j]; g_ref[m] += r * hs[tau, j]
    assert rel_err(sd[:, :H].cpu().numpy(), s_ref) < 2e-6
    assert rel_err(gd[:, :H].cpu().numpy(), g_ref) < 2e-6
    # final-value collection
    sel = np.minimum(sf, T_ - 1).astype(np.int32)
    out = torch.empty((NM, ld), device="cuda")
    K.select_slab_rows(out.data_ptr(), hsd.data_ptr(), NM * ld, dev(sel).data_ptr(), NM, H, ld)
    torch.cuda.synchronize()
    assert np.array_equal(out[:, :H].cpu().numpy(), hs[sel, np.arange(NM)])


@pytest.mark.parametrize("M0,M1", [(200, 200), (300, 77), (3000, 3000), (900, 5000), (20000, 20000), (0, 100)])
def test_gemm_nt_pair_equals_two_calls(K, M0, M1):
    """The paired launch (accumulate-in-place product + multiply-accumulate product, the pair of the
    tree GRU's backward pass) gives exactly what two separate calls give, at every size class."""
    rs = np.random.RandomState(M0 + M1)
    H, ld = 450, 452
    Wa = dev_rows(rs.standard_normal((H, H)).astype(np.float32) * 0.1, ld)
    Wb = dev_rows(rs.standard_normal((H, H)).astype(np.float32) * 0.1, ld)
    NM = max(M0, M1, 1) + 50
    rows0 = dev(np.sort(rs.choice(NM, size=M0, replace=False)).astype(np.int32)) if M0 else None
    rows1 = dev(np.sort(rs.choice(NM, size=M1, replace=False)).astype(np.int32)) if M1 else None
    A0 = dev_rows(rs.standard_normal((max(M0, 1), H)).astype(np.float32), ld)
    A1 = dev_rows(rs.standard_normal((max(M1, 1), H)).astype(np.float32), ld)
    mul = dev_rows(rs.standard_normal((NM, H)).astype(np.float32), ld)
    out = {}
    for how in ("pair", "two"):
        ds = dev_rows(np.ones((NM, H), np.float32), ld)
        dg = dev_rows(np.zeros((NM, H), np.float32), ld)
        acc = dev_rows(np.full((NM, H), 0.5, np.float32), ld)
        a0 = dict(add=ds.data_ptr(), ld_add=ld, c_rows=rows0.data_ptr() if M0 else None)
        a1 = dict(mode=K.EPI_MUL_ACC, aux0=mul.data_ptr(), out2=acc.data_ptr(), ld_aux=ld,
                  c_rows=rows1.data_ptr() if M1 else None)
        if how == "pair":
            K.gemm_nt_pair(K.nt_problem(A0.data_ptr(), ld, Wa.data_ptr(), ld, ds.data_ptr(), ld, M0, H, H, **a0),
                           K.nt_problem(A1.data_ptr(), ld, Wb.data_ptr(), ld, dg.data_ptr(), ld, M1, H, H, **a1))
        else:
            K.gemm_nt(A0.data_ptr(), ld, Wa.data_ptr(), ld, ds.data_ptr(), ld, M0, H, H, **a0)
            K.gemm_nt(A1.data_ptr(), ld, Wb.data_ptr(), ld, dg.data_ptr(), ld, M1, H, H, **a1)
        torch.cuda.synchronize()
        out[how] = (ds.clone(), dg.clone(), acc.clone())
    for x, y in zip(out["pair"], out["two"]):
        assert torch.equal(x, y)
    if M0:
        assert float((out["pair"][0] - 1.0).abs().max()) > 0.0  # the first product did land


@pytest.mark.parametrize("n_rows,with_rows", [(5, True), (300, False), (513, True), (4100, True), (12000, False)])
def test_gru_update_equals_the_two_products_it_replaces(K, n_rows, with_rows):
    """z, p and the blended state of one GRU level from one launch, bit for bit what two gemm_nt calls
    (sigmoid epilogue, then the blend epilogue) produce, for the small, the tiled and the general route."""
    rs = np.random.RandomState(n_rows)
    H, ld, V = 450, 452, 37
    NM = n_rows + 9
    def buf(n, scale=1.0):
        a = np.zeros((n, ld), np.float32)
        a[:, :H] = rs.standard_normal((n, H)).astype(np.float32) * scale
        return torch.from_numpy(a).cuda()
    S, G = buf(NM, 0.5), buf(NM, 0.5)
    Wz, Wh = buf(H, 0.05), buf(H, 0.05)
    Pz, Ph = buf(V), buf(V)
    wid = torch.from_numpy(rs.randint(0, V, size=NM).astype(np.int32)).cuda()
    rows = None
    if with_rows:
        r = rs.permutation(NM)[:n_rows].astype(np.int32)
        r[0] = 0  # the padding message among the rows: its new state must stay zero
        rows = torch.from_numpy(np.sort(r)).cuda()
    out = []
    for fused in (True, False):
        Z, P, Hn = (torch.full((NM, ld), 7.0, device="cuda") for _ in range(3))
        if fused:
            K.gru_update(S.data_ptr(), G.data_ptr(), ld, Wz.data_ptr(), Wh.data_ptr(), ld, Pz.data_ptr(), Ph.data_ptr(),
                         ld, wid.data_ptr(), K.ptr(rows), n_rows, Z.data_ptr(), P.data_ptr(), Hn.data_ptr(), H, 0)
        else:
            K.gemm_nt(S.data_ptr(), ld, Wz.data_ptr(), ld, Z.data_ptr(), ld, n_rows, H, H, add=Pz.data_ptr(), ld_add=ld,
                      add_idx=wid.data_ptr(), act=K.ACT_SIGMOID, a_rows=K.ptr(rows), c_rows=K.ptr(rows))
            K.gemm_nt(G.data_ptr(), ld, Wh.data_ptr(), ld, P.data_ptr(), ld, n_rows, H, H, add=Ph.data_ptr(), ld_add=ld,
                      add_idx=wid.data_ptr(), mode=K.EPI_GRU_BLEND, aux0=Z.data_ptr(), aux1=S.data_ptr(),
                      out2=Hn.data_ptr(), ld_aux=ld, a_rows=K.ptr(rows), c_rows=K.ptr(rows), zero_row=0)
        torch.cuda.synchronize()
        out.append((Z.cpu().numpy(), P.cpu().numpy(), Hn.cpu().numpy()))
    for a, b, what in zip(out[0], out[1], "zph"):
        assert np.array_equal(a, b), what
    sel = rows.cpu().numpy() if with_rows else np.arange(n_rows)
    Sn, Gn = S.cpu().numpy().astype(np.float64)[:, :H], G.cpu().numpy().astype(np.float64)[:, :H]
    w = wid.cpu().numpy()
    z = 1 / (1 + np.exp(-(Sn[sel] @ Wz.cpu().numpy().astype(np.float64)[:, :H].T + Pz.cpu().numpy()[w[sel]][:, :H])))
    p = np.tanh(Gn[sel] @ Wh.cpu().numpy().astype(np.float64)[:, :H].T + Ph.cpu().numpy()[w[sel]][:, :H])
    h = (1 - z) * Sn[sel] + z * p
    h[sel == 0] = 0
    assert rel_err(out[0][0][sel][:, :H], z) < 1e-5 and rel_err(out[0][1][sel][:, :H], p) < 1e-5
    assert rel_err(out[0][2][sel][:, :H], h) < 1e-5
    assert not out[0][2][sel][:, H:].any()  # padding columns written as zero
    untouched = np.setdiff1d(np.arange(NM), sel)
    assert (out[0][2][untouched] == 7.0).all()  # rows outside the list are not written


@pytest.mark.parametrize("n_rows", [3, 300, 513, 4100, 12000])
def test_gru_bwd_dh_equals_the_product_and_the_pointwise_kernel_it_replaces(K, n_rows):
    """dh of a backward level and the pre-activation gradients that follow from it in one launch, bit for bit what
    gemm_nt (add epilogue, scattered output rows) followed by gru_bwd_pointwise produces - for the small, the
    tiled and the general route; the padding message among the rows, rows whose state is that of iteration 0
    (no s), compact dz / dp and ds by message id."""
    rs = np.random.RandomState(n_rows)
    H, ld, T = 450, 452, 20
    NM = n_rows + 11
    def buf(n, scale=1.0):
        a = np.zeros((n, ld), np.float32)
        a[:, :H] = rs.standard_normal((n, H)).astype(np.float32) * scale
        return torch.from_numpy(a).cuda()
    du, UrT, dhdir = buf(n_rows, 0.5), buf(H, 0.05), buf(NM, 0.5)
    z, p, s = buf(NM), buf(NM), buf(NM)
    z, p = torch.sigmoid(z), torch.tanh(p)
    z[:, H:] = 0; p[:, H:] = 0
    r = np.sort(rs.permutation(NM)[:n_rows]).astype(np.int32)
    r[0] = 0  # the padding message: its gradient is dropped
    rows = torch.from_numpy(r).cuda()
    sf = rs.randint(1, T + 3, size=NM).astype(np.int32)  # sf == 1: the row's state is that of iteration 0
    sf_d = torch.from_numpy(sf).cuda()
    out = []
    for fused in (True, False):
        dh, ds = (torch.full((NM, ld), 7.0, device="cuda") for _ in range(2))
        dz, dp = (torch.full((n_rows, ld), 7.0, device="cuda") for _ in range(2))
        if fused:
            K.gru_bwd_dh(du.data_ptr(), UrT.data_ptr(), ld, dhdir.data_ptr(), dh.data_ptr(), rows.data_ptr(), n_rows,
                         z.data_ptr(), p.data_ptr(), s.data_ptr(), 0, T, sf_d.data_ptr(), dz.data_ptr(), dp.data_ptr(),
                         ds.data_ptr(), H, ld, 0)
        else:
            K.gemm_nt(du.data_ptr(), ld, UrT.data_ptr(), ld, dh.data_ptr(), ld, n_rows, H, H, add=dhdir.data_ptr(),
                      ld_add=ld, c_rows=rows.data_ptr())
            K.gru_bwd_pointwise(dh.data_ptr(), z.data_ptr(), p.data_ptr(), s.data_ptr(), 0, T, sf_d.data_ptr(),
                                rows.data_ptr(), n_rows, dz.data_ptr(), dp.data_ptr(), ds.data_ptr(), H, ld, 0)
        torch.cuda.synchronize()
        out.append([t.cpu().numpy() for t in (dh, dz, dp, ds)])
    for a, b, what in zip(out[0], out[1], ("dh", "dz", "dp", "ds")):
        assert np.array_equal(a, b), what
    dh64 = du.cpu().numpy().astype(np.float64)[:, :H] @ UrT.cpu().numpy().astype(np.float64)[:, :H].T \
        + dhdir.cpu().numpy()[r][:, :H]
    assert rel_err(out[0][0][r][:, :H], dh64) < 1e-5
    d = dh64.copy()
    d[r == 0] = 0
    zz, pp = z.cpu().numpy().astype(np.float64)[r][:, :H], p.cpu().numpy().astype(np.float64)[r][:, :H]
    ss = np.where((np.minimum(T, sf[r] - 1) > 0)[:, None], s.cpu().numpy().astype(np.float64)[r][:, :H], 0.0)
    assert rel_err(out[0][1][:, :H], d * (pp - ss) * zz * (1 - zz)) < 1e-5
    assert rel_err(out[0][2][:, :H], d * zz * (1 - pp * pp)) < 1e-5
    assert rel_err(out[0][3][r][:, :H], d * (1 - zz)) < 1e-5
    assert not out[0][1][:, H:].any() and not out[0][2][:, H:].any() and not out[0][3][r][:, H:].any()
    untouched = np.setdiff1d(np.arange(NM), r)
    assert (out[0][0][untouched] == 7.0).all() and (out[0][3][untouched] == 7.0).all()
    assert K.lib.jtvae_gru_bwd_dh(None, UrT.data_ptr(), ld, dhdir.data_ptr(), dh.data_ptr(), None, 4, z.data_ptr(),
                                  p.data_ptr(), s.data_ptr(), 0, T, None, dz.data_ptr(), dp.data_ptr(), ds.data_ptr(),
                                  H, ld, 0, None) == K.E_NULL
    assert K.lib.jtvae_gru_bwd_dh(du.data_ptr(), UrT.data_ptr(), ld, dhdir.data_ptr(), dh.data_ptr(), None, 4,
                                  z.data_ptr(), p.data_ptr(), s.data_ptr(), 0, T, None, dz.data_ptr(), dp.data_ptr(),
                                  ds.data_ptr(), H, 448, 0, None) == K.E_SHAPE


def test_level_sweeps_reject_bad_arguments_and_accept_an_empty_schedule(K):
    """jtvae_tree_gru_forward / _backward: argument errors come back before any launch; a schedule without
    levels (every message settled at iteration 0) is a no-op."""
    import ctypes as C

    assert K.lib.jtvae_tree_gru_forward(None, None) == K.E_NULL
    assert K.lib.jtvae_tree_gru_backward(None, None) == K.E_NULL
    f = K.TreeFwd(H=450, ld=452, n_levels=0)
    assert K.lib.jtvae_tree_gru_forward(C.byref(f), K.stream_ptr()) == 0
    b = K.TreeBwd(H=450, ld=452, n_levels=0, T=20)
    assert K.lib.jtvae_tree_gru_backward(C.byref(b), K.stream_ptr()) == 0
    f = K.TreeFwd(H=450, ld=448, n_levels=0)  # leading dimension shorter than a row
    assert K.lib.jtvae_tree_gru_forward(C.byref(f), K.stream_ptr()) == K.E_SHAPE
    f = K.TreeFwd(H=450, ld=452, n_levels=2)  # levels announced, no row lists
    assert K.lib.jtvae_tree_gru_forward(C.byref(f), K.stream_ptr()) == K.E_NULL
    b = K.TreeBwd(H=450, ld=452, n_levels=2, T=20)
    assert K.lib.jtvae_tree_gru_backward(C.byref(b), K.stream_ptr()) == K.E_NULL
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_rows,F,H", [(1, 51, 450), (777, 44, 450), (40000, 39, 450), (300, 12, 96)])
def test_onehot_linear_forward_and_weight_gradient(n_rows, F, H):
    """jtvae_onehot_linear / _bwd against fp64 numpy on byte rows with a few set entries (and the odd entry > 1),
    and bitwise against themselves on a second run (fixed summation order)."""
    from fast_jtnn_hip import _capi as K

    rs = np.random.RandomState(n_rows + F)
    ld = K.round_ld(H)
    feats = (rs.rand(n_rows, F) < 0.15).astype(np.uint8)
    feats[rs.randint(0, n_rows, size=3), rs.randint(0, F, size=3)] = 3
    W = rs.standard_normal((H, F)).astype(np.float32)
    bias = rs.standard_normal(H).astype(np.float32)
    Wt = torch.zeros((F, ld), dtype=torch.float32, device="cuda")
    Wt[:, :H] = torch.from_numpy(W.T.copy()).cuda()
    f_dev = torch.from_numpy(feats).cuda()
    b_dev = torch.from_numpy(bias).cuda()
    outs = []
    for _ in range(2):
        out = torch.full((n_rows, ld), float("nan"), device="cuda")
        K.onehot_linear(f_dev.data_ptr(), n_rows, F, Wt.data_ptr(), ld, b_dev.data_ptr(), out.data_ptr(), ld, H)
        outs.append(out.cpu().numpy())
    want = feats.astype(np.float64) @ W.T.astype(np.float64) + bias
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0][:, :H] - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)
    assert not outs[0][:, H:].any()  # padding columns written as zero
    # weight gradient, fresh and accumulated
    d = torch.zeros((n_rows, ld), device="cuda")
    d[:, :H] = torch.from_numpy(rs.standard_normal((n_rows, H)).astype(np.float32)).cuda()
    want_dw = d.cpu().numpy()[:, :H].astype(np.float64).T @ feats.astype(np.float64)
    init = torch.from_numpy(rs.standard_normal((H, F)).astype(np.float32)).cuda()
    grads = []
    for acc in (False, False, True):
        dW = init.clone()
        K.onehot_linear_bwd(f_dev.data_ptr(), n_rows, F, d.data_ptr(), ld, H, dW.data_ptr(), F, acc, "cuda")
        grads.append(dW.cpu().numpy())
    assert np.array_equal(grads[0], grads[1])
    scale = max(np.abs(want_dw).max(), 1.0)
    assert np.abs(grads[0] - want_dw).max() <= 2e-5 * scale
    assert np.abs(grads[2] - (want_dw + init.cpu().numpy())).max() <= 2e-5 * scale
    assert K.lib.jtvae_onehot_linear(16, 4, 400, 16, 452, None, 16, 452, 450, None) == K.E_WORKSPACE  # table too big


@pytest.mark.gpu
@pytest.mark.parametrize("M", [10001, 20480, 33333])
def test_large_products_in_split_form_are_fp32_accurate(K, M):
    """Products of more than ~10 000 rows run on the bf16 matrix instruction with operands split into three bf16
    parts (jtvae_set_split_products).  Against fp64 they must be as accurate as the fp32 instruction's fma chain:
    both within 2e-6 of sum |a w| per element, the split form's rms error within 1.5x of the fp32 form's; gathered /
    scattered rows, bias, addend and activation behave as in the fp32 form; padding columns stay zero."""
    import torch
    H, ld = 450, 452
    g = torch.Generator(device="cuda").manual_seed(M)
    A = torch.zeros(M, ld, device="cuda")
    A[:, :H] = torch.relu(torch.randn(M, H, device="cuda", generator=g)) * torch.exp(torch.randn(M, H, device="cuda", generator=g))
    A[:, H:] = float("nan")  # padding columns of an input may hold anything
    W = torch.zeros(H, ld, device="cuda")
    W[:, :H] = 0.05 * torch.randn(H, H, device="cuda", generator=g)
    ref = A[:, :H].double() @ W[:, :H].double().t()
    mag = A[:, :H].double().abs() @ W[:, :H].double().abs().t()
    out = {}
    for on in (True, False):
        was = K.set_split_products(on)
        try:
            C_ = torch.full((M, ld), 7.0, device="cuda")
            K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C_.data_ptr(), ld, M, H, H)
            torch.cuda.synchronize()
        finally:
            K.set_split_products(was)
        assert bool((C_[:, H:] == 0).all())
        err = (C_[:, :H].double() - ref).abs()
        assert float((err / mag).max()) < 2e-6, on
        out[on] = (C_, float(err.pow(2).mean().sqrt()))
    assert out[True][1] < 1.5 * out[False][1]
    assert not torch.equal(out[True][0], out[False][0])  # two forms, not one kernel under two names
    # row lists + bias + addend + activation through the split form
    rows = torch.randperm(M, device="cuda", generator=g)[: M - 37].to(torch.int32).contiguous()
    bias = torch.randn(H, device="cuda", generator=g)
    add = torch.zeros(M, ld, device="cuda")
    add[:, :H] = torch.randn(M, H, device="cuda", generator=g)
    C_ = torch.zeros(M, ld, device="cuda")
    K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C_.data_ptr(), ld, rows.numel(), H, H, bias=bias.data_ptr(),
              add=add.data_ptr(), ld_add=ld, act=K.ACT_TANH, a_rows=rows.data_ptr(), c_rows=rows.data_ptr())
    torch.cuda.synchronize()
    r = rows.long()
    want = torch.tanh(ref[r] + bias.double() + add[r, :H].double())
    assert float((C_[r, :H].double() - want).abs().max()) < 2e-5
    untouched = torch.ones(M, dtype=torch.bool, device="cuda")
    untouched[r] = False
    assert bool((C_[untouched] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("M", [4096, 40898, 100003])
def test_weight_gradient_in_split_form_is_fp32_accurate(K, M):
    """jtvae_gemm_tn over >= 4096 rows in the split form (transposing LDS reads feed the bf16 instruction): against
    fp64 as accurate as the fp32 form to a factor of 2.5 in rms, bias column and row-pair lists included."""
    import torch
    H, ld = 450, 452
    g = torch.Generator(device="cuda").manual_seed(M)
    X = torch.zeros(M, ld, device="cuda")
    X[:, :H] = torch.randn(M, H, device="cuda", generator=g) * (torch.rand(M, H, device="cuda", generator=g) < 0.5)
    Y = torch.zeros(M, ld, device="cuda")
    Y[:, :H] = torch.relu(torch.randn(M, H, device="cuda", generator=g))
    xr = torch.randint(0, M, (M,), device="cuda", dtype=torch.int32, generator=g)
    yr = torch.randint(0, M, (M,), device="cuda", dtype=torch.int32, generator=g)
    ref = X[:, :H].double().t() @ Y[:, :H].double()
    ref_b = X[:, :H].double().sum(0)
    ref_p = X[xr.long(), :H].double().t() @ Y[yr.long(), :H].double()
    rms = {}
    for on in (True, False):
        was = K.set_split_products(on)
        try:
            dW = torch.empty(H, H, device="cuda")
            db = torch.empty(H, device="cuda")
            K.gemm_tn(X.data_ptr(), ld, Y.data_ptr(), ld, dW.data_ptr(), H, M, H, H, False, "cuda", dbias=db.data_ptr())
            dP = torch.empty(H, H, device="cuda")
            K.gemm_tn(X.data_ptr(), ld, Y.data_ptr(), ld, dP.data_ptr(), H, M, H, H, False, "cuda",
                      x_rows=xr.data_ptr(), y_rows=yr.data_ptr())
            torch.cuda.synchronize()
        finally:
            K.set_split_products(was)
        e = float((dW.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        eb = float((db.double() - ref_b).abs().max() / ref_b.abs().max())
        ep = float((dP.double() - ref_p).pow(2).mean().sqrt() / ref_p.pow(2).mean().sqrt())
        assert e < 2e-6 and eb < 1e-6 and ep < 2e-6, (on, e, eb, ep)
        rms[on] = e
    assert rms[True] < 2.5 * rms[False]


@pytest.mark.gpu
def test_graph_mpn_sweeps_reject_bad_arguments(K):
    """jtvae_graph_mpn_forward / _backward check their argument blocks on the host before any launch."""
    import ctypes as C
    assert K.lib.jtvae_graph_mpn_forward(None, None) == -1  # JTVAE_E_NULL
    assert K.lib.jtvae_graph_mpn_backward(None, None) == -1
    a = K.MpnFwdArgs()
    a.depth, a.H, a.ld = 0, 450, 452
    assert K.lib.jtvae_graph_mpn_forward(C.byref(a), None) == -2  # JTVAE_E_SHAPE: depth < 1
    a.depth, a.H, a.ld = 3, 450, 448
    assert K.lib.jtvae_graph_mpn_forward(C.byref(a), None) == -2  # ld < H
    a.ld = 452
    assert K.lib.jtvae_graph_mpn_forward(C.byref(a), None) == -1  # no message buffers
    b = K.MpnBwdArgs()
    b.depth, b.H, b.ld = 3, 450, 452
    assert K.lib.jtvae_graph_mpn_backward(C.byref(b), None) == -1
    b.n_bonds = -1
    assert K.lib.jtvae_graph_mpn_backward(C.byref(b), None) == -2
    # jtvae_tree_encoder_backward: no block, no sweep block, inconsistent sizes, missing buffers
    assert K.lib.jtvae_tree_encoder_backward(None, None, None) == -1
    t = K.TreeEncBwd()
    assert K.lib.jtvae_tree_encoder_backward(C.byref(t), None, None) == -1
    sweep = K.TreeBwd()
    t.sweep = C.pointer(sweep)
    t.H, t.ld = 450, 448
    assert K.lib.jtvae_tree_encoder_backward(C.byref(t), None, None) == -2
    t.ld = 452
    t.NM = t.B = t.NW = t.V = t.n_z = t.n_u = t.n_e = 4
    assert K.lib.jtvae_tree_encoder_backward(C.byref(t), None, None) == -1
    assert K.lib.jtvae_tree_encoder_forward(None, None, None) == -1
    f = K.TreeEncFwd()
    assert K.lib.jtvae_tree_encoder_forward(C.byref(f), None, None) == -1  # no sweep block
    fs = K.TreeFwd()
    f.sweep = C.pointer(fs)
    f.NM = f.B = f.NW = f.V = 4
    f.H, f.ld = 450, 452
    assert K.lib.jtvae_tree_encoder_forward(C.byref(f), None, None) == -2  # the sweep block disagrees on H / ld
    fs.H, fs.ld = 450, 452
    assert K.lib.jtvae_tree_encoder_forward(C.byref(f), None, None) == -1  # no buffers


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 77, 512])
def test_small_tile_row_limit_picks_the_tiling_and_restores(K, M):
    """jtvae_set_small_tile_rows: a product of <= 512 rows takes the one-tile kernel by default and the 128 x 32
    tiling with the limit at 0; both are fp32 fma chains, each within 2e-6 of sum |a w| of the fp64 product.  The
    setting is per thread, returns the previous value, and -1 restores the default."""
    import threading
    import torch
    H, ld = 450, 452
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    A = torch.zeros(M, ld, device="cuda")
    A[:, :H] = torch.randn(M, H, device="cuda", generator=g)
    W = torch.zeros(H, ld, device="cuda")
    W[:, :H] = 0.05 * torch.randn(H, H, device="cuda", generator=g)
    ref = A[:, :H].double() @ W[:, :H].double().t()
    mag = A[:, :H].double().abs() @ W[:, :H].double().abs().t()

    def run():
        C_ = torch.full((M, ld), 3.0, device="cuda")
        K.gemm_nt(A.data_ptr(), ld, W.data_ptr(), ld, C_.data_ptr(), ld, M, H, H)
        torch.cuda.synchronize()
        return C_

    assert K.lib.jtvae_set_small_tile_rows(-1) == -1
    one_tile = run()
    with K.small_tile_rows(0):
        narrow = run()
        seen = []
        t = threading.Thread(target=lambda: seen.append(K.lib.jtvae_set_small_tile_rows(-1)))
        t.start()
        t.join()
        assert seen == [-1]  # another thread still has the default
        with K.small_tile_rows(None):  # None: leave as is
            assert K.lib.jtvae_set_small_tile_rows(0) == 0
    assert K.lib.jtvae_set_small_tile_rows(-1) == -1  # restored on exit
    assert K.lib.jtvae_set_small_tile_rows(-7) == -1  # any negative value is "default"
    assert torch.equal(run(), one_tile)
    for C_ in (one_tile, narrow):
        assert bool((C_[:, H:] == 0).all())
        assert float(((C_[:, :H].double() - ref).abs() / mag).max()) < 2e-6
