"""GPU: the allocation-free weight-gradient entry point (jtvae_gemm_tn_static) against the entry point
it is a variant of (jtvae_gemm_tn), through the C ABI, on two calls in a row, the second smaller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(K, M, N, Kd, seed, with_rows, with_bias, accumulate):
    rs = np.random.RandomState(seed)
    ld = K.round_ld(max(N, Kd))
    rows_x = M + 7
    X = torch.from_numpy(rs.standard_normal((rows_x, ld)).astype(np.float32)).cuda()
    Y = torch.from_numpy(rs.standard_normal((rows_x, ld)).astype(np.float32)).cuda()
    xr = yr = None
    if with_rows:
        xr = torch.from_numpy(rs.randint(0, rows_x, size=M).astype(np.int32)).cuda()
        yr = torch.from_numpy(rs.randint(0, rows_x, size=M).astype(np.int32)).cuda()
    init = torch.from_numpy(rs.standard_normal((N, Kd)).astype(np.float32)).cuda()
    binit = torch.from_numpy(rs.standard_normal((N,)).astype(np.float32)).cuda()
    out = []
    for static in (False, True):
        dW, db = init.clone(), binit.clone()
        args = (X.data_ptr(), ld, K.ptr(xr), Y.data_ptr(), ld, K.ptr(yr), dW.data_ptr(), Kd,
                db.data_ptr() if with_bias else None, M, N, Kd, int(accumulate))
        if static:
            rc = K.lib.jtvae_gemm_tn_static(*args, K.stream_ptr())
        else:
            nbytes = K.lib.jtvae_gemm_tn_workspace_bytes(M, N, Kd)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
            rc = K.lib.jtvae_gemm_tn(*args, ws.data_ptr(), nbytes, K.stream_ptr())
        K.check(rc, "gemm_tn%s" % ("_static" if static else ""))
        torch.cuda.synchronize()
        out.append((dW.cpu().numpy(), db.cpu().numpy()))
    # fp64 statement of the contraction, as the sanity anchor of both
    Xn, Yn = X.cpu().numpy().astype(np.float64), Y.cpu().numpy().astype(np.float64)
    xi = xr.cpu().numpy() if with_rows else np.arange(M)
    yi = yr.cpu().numpy() if with_rows else np.arange(M)
    want = Xn[xi][:, :N].T @ Yn[yi][:, :Kd] + (init.cpu().numpy() if accumulate else 0.0)
    return out, want


def test_static_scratch_weight_gradient_equals_workspace_form_on_two_calls_in_a_row():
    from fast_jtnn_hip import _capi as K

    assert K.lib.jtvae_gemm_tn_static_max_bytes() == 64 << 20
    # first call: the largest slab plan of the path (64 slabs); second call: a smaller product, which
    # finds the first call's partials still in the array
    for M, N, Kd, rows, bias, acc in ((85181, 450, 450, False, True, False), (3001, 450, 39, True, False, True),
                                      (17, 450, 450, True, True, False)):
        (a, b), want = _both(K, M, N, Kd, M, rows, bias, acc)
        assert np.array_equal(a[0], b[0]), (M, N, Kd)  # same slab plan: bit-identical
        if bias:
            assert np.array_equal(a[1], b[1])
        scale = np.abs(want).max()
        assert np.abs(b[0] - want).max() <= 1e-4 * scale  # fp32 against fp64, the path's 1e-4 bound


def test_static_scratch_rejects_a_product_beyond_its_array():
    from fast_jtnn_hip import _capi as K

    x = torch.zeros((64, 4096), device="cuda")
    dW = torch.zeros((4096, 4096), device="cuda")
    M = 1 << 20  # many slabs of 4096 x 4096 partials: far beyond 64 MiB; rejected before any launch
    assert K.lib.jtvae_gemm_tn_workspace_bytes(M, 4096, 4096) > K.lib.jtvae_gemm_tn_static_max_bytes()
    rc = K.lib.jtvae_gemm_tn_static(x.data_ptr(), 4096, None, x.data_ptr(), 4096, None, dW.data_ptr(), 4096, None,
                                    M, 4096, 4096, 0, K.stream_ptr())
    assert rc == K.E_WORKSPACE


def test_static_scratch_follows_the_device_of_the_stream():
    """The array is per GPU: a call whose stream (and operands) belong to another GPU than the current one must
    use that GPU's array.  Needs two visible GPUs."""
    from fast_jtnn_hip import _capi as K

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    rs = np.random.RandomState(3)
    M, N, Kd = 5000, 450, 450
    ld = K.round_ld(N)
    with torch.cuda.device(1):
        X = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32)).cuda(1)
        Y = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32)).cuda(1)
        dW = torch.zeros((N, Kd), device="cuda:1")
        stream = torch.cuda.Stream(device=1)
        torch.cuda.synchronize()
    # current device 0, stream of device 1
    rc = K.lib.jtvae_gemm_tn_static(X.data_ptr(), ld, None, Y.data_ptr(), ld, None, dW.data_ptr(), Kd, None, M, N, Kd, 0,
                                    stream.cuda_stream)
    K.check(rc, "gemm_tn_static on the second GPU")
    stream.synchronize()
    want = X.cpu().numpy().astype(np.float64)[:, :N].T @ Y.cpu().numpy().astype(np.float64)[:, :Kd]
    assert np.abs(dW.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
