"""Dense layers of the model's heads on the package's own products.

``nn.Linear`` on this platform goes through the BLAS library: ~0.1-0.6 ms of host time per call and, for the fp32
[n, 450..900] x [450, .] shapes of the teacher-forced decoder's heads (fast_jtnn/jtnn_dec.py:42-53, 146-193), a
quarter of the rate ``jtvae_gemm_nt`` / ``jtvae_gemm_tn`` reach on the same shapes.  The two functions here keep the
``nn.Linear`` modules as the owners of the parameters (state dicts stay the reference's) and replace the calls:

* ``linear(x, lin)``: ``x W^T + b``.
* ``table_linear(h, src, idx, lin, h_first, relu)``: ``act(lin(cat([h, src[idx]], 1)))`` (``h_first``) or
  ``act(lin(cat([src[idx], h], 1)))`` without forming the concatenation or the gathered rows: the few rows of
  ``src`` (one per tree, or one per vocabulary word) go through their columns of the weight once, and the product
  over the ``h`` rows adds the row that ``idx`` names in its epilogue, bias and ReLU included.
"""
from __future__ import annotations

import torch

from . import _capi as K
from .mpn import as_row_buffer, dense_view, pack_weights


def _need_cuda(t, what):
    if t.device.type != "cuda":
        raise RuntimeError("fast_jtnn_hip.%s runs on an MI355X only (input is on %s); there is no CPU path"
                           % (what, t.device))


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        dev = x.device
        n, kin = x.shape
        out_f = W.shape[0]
        ldx, ldo = K.round_ld(kin), K.round_ld(out_f)
        with torch.cuda.device(dev):
            xp = as_row_buffer(x, kin, ldx)
            (Wp,) = pack_weights([(W, 0, kin, False)])
            y = K.rows(n, ldo, dev)
            K.gemm_nt(xp.data_ptr(), ldx, Wp.data_ptr(), ldx, y.data_ptr(), ldo, n, out_f, kin,
                      bias=None if b is None else b.detach().data_ptr())
        if any(ctx.needs_input_grad):
            ctx.keep = (xp, W, kin, out_f, ldx, ldo, b is not None)
        return dense_view(y, out_f)

    @staticmethod
    def backward(ctx, dy):
        xp, W, kin, out_f, ldx, ldo, has_b = ctx.keep
        dev, n = xp.device, xp.shape[0]
        dx = dW = db = None
        with torch.cuda.device(dev):
            dyp = as_row_buffer(dy, out_f, ldo)
            if ctx.needs_input_grad[0]:
                (WT,) = pack_weights([(W, 0, kin, True)])  # [kin, ldo]
                dxp = K.rows(n, ldx, dev)
                K.gemm_nt(dyp.data_ptr(), ldo, WT.data_ptr(), ldo, dxp.data_ptr(), ldx, n, kin, out_f)
                dx = dense_view(dxp, kin)
            if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
                dW = torch.empty((out_f, kin), dtype=torch.float32, device=dev)
                db = torch.empty((out_f,), dtype=torch.float32, device=dev) if has_b else None
                K.gemm_tn(dyp.data_ptr(), ldo, xp.data_ptr(), ldx, dW.data_ptr(), kin, n, out_f, kin, False, dev,
                          dbias=None if db is None else db.data_ptr())
        return dx, dW, db


def linear(x: torch.Tensor, lin: torch.nn.Linear) -> torch.Tensor:
    """``lin(x)`` for a 2-d fp32 ``x`` on the GPU."""
    _need_cuda(x, "linear")
    return _Linear.apply(x, lin.weight, lin.bias)


class _TableLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, src, idx32, W, b, h_first, relu):
        dev = h.device
        (n, nh), (m, ns) = h.shape, src.shape
        out_f = W.shape[0]
        if W.shape[1] != nh + ns or idx32.shape[0] != n or idx32.dtype != torch.int32:
            raise ValueError("table_linear: weight %s against inputs of %d + %d columns, %d indices (%s) for %d rows"
                             % (tuple(W.shape), nh, ns, idx32.shape[0], idx32.dtype, n))
        h_col0, s_col0 = (0, nh) if h_first else (ns, 0)
        ldh, lds, ldo = K.round_ld(nh), K.round_ld(ns), K.round_ld(out_f)
        with torch.cuda.device(dev):
            hp, sp = as_row_buffer(h, nh, ldh), as_row_buffer(src, ns, lds)
            Wh, Ws = pack_weights([(W, h_col0, nh, False), (W, s_col0, ns, False)])
            table = K.rows(m, ldo, dev)
            K.gemm_nt(sp.data_ptr(), lds, Ws.data_ptr(), lds, table.data_ptr(), ldo, m, out_f, ns,
                      bias=None if b is None else b.detach().data_ptr())
            y = K.rows(n, ldo, dev)
            K.gemm_nt(hp.data_ptr(), ldh, Wh.data_ptr(), ldh, y.data_ptr(), ldo, n, out_f, nh, add=table.data_ptr(),
                      ld_add=ldo, add_idx=idx32.data_ptr(), act=K.ACT_RELU if relu else K.ACT_NONE)
        if any(ctx.needs_input_grad):
            ctx.keep = (hp, sp, idx32, W, y if relu else None, (nh, ns, out_f, h_col0, s_col0, ldh, lds, ldo),
                        b is not None)
        return dense_view(y, out_f)

    @staticmethod
    def backward(ctx, dy):
        hp, sp, idx32, W, y, (nh, ns, out_f, h_col0, s_col0, ldh, lds, ldo), has_b = ctx.keep
        dev, n, m = hp.device, hp.shape[0], sp.shape[0]
        dh = dsrc = dW = db = None
        with torch.cuda.device(dev):
            if y is None:
                dpre = as_row_buffer(dy, out_f, ldo)
            else:
                dpre = K.rows(n, ldo, dev)
                torch.mul(dy, y[:, :out_f] > 0, out=dpre[:, :out_f])
            # gradient of the table: rows of dpre summed by the table row they read (order of summation is the
            # hardware's: equal up to rounding from run to run)
            dtable = K.rows(m, ldo, dev, zero=True)
            dtable[:, :out_f].index_add_(0, idx32.long(), dpre[:, :out_f])
            need_w = ctx.needs_input_grad[3] or (has_b and ctx.needs_input_grad[4])
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                WhT, WsT = pack_weights([(W, h_col0, nh, True), (W, s_col0, ns, True)])  # [nh, ldo], [ns, ldo]
                if ctx.needs_input_grad[0]:
                    dhp = K.rows(n, ldh, dev)
                    K.gemm_nt(dpre.data_ptr(), ldo, WhT.data_ptr(), ldo, dhp.data_ptr(), ldh, n, nh, out_f)
                    dh = dense_view(dhp, nh)
                if ctx.needs_input_grad[1]:
                    dsp = K.rows(m, lds, dev)
                    K.gemm_nt(dtable.data_ptr(), ldo, WsT.data_ptr(), ldo, dsp.data_ptr(), lds, m, ns, out_f)
                    dsrc = dense_view(dsp, ns)
            if need_w:
                kin = nh + ns
                dW = torch.empty((out_f, kin), dtype=torch.float32, device=dev)
                db = torch.empty((out_f,), dtype=torch.float32, device=dev) if has_b else None
                K.gemm_tn(dpre.data_ptr(), ldo, hp.data_ptr(), ldh, K.ptr(dW, h_col0), kin, n, out_f, nh, False, dev)
                K.gemm_tn(dtable.data_ptr(), ldo, sp.data_ptr(), lds, K.ptr(dW, s_col0), kin, m, out_f, ns, False, dev,
                          dbias=None if db is None else db.data_ptr())
        return dh, dsrc, None, dW, db, None, None


def table_linear(h, src, idx32, lin, h_first=True, relu=False):
    """``act(lin(cat([h, src[idx32]], 1)))`` (``h_first``) or ``act(lin(cat([src[idx32], h], 1)))``; ``idx32`` int32
    on the device, one entry per row of ``h``."""
    _need_cuda(h, "table_linear")
    return _TableLinear.apply(h, src, idx32, lin.weight, lin.bias, bool(h_first), bool(relu))
