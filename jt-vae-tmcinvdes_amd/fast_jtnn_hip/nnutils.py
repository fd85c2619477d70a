"""Tensor helpers with the reference's names (fast_jtnn/nnutils.py:6-18).

``create_var`` in the reference wraps a tensor and calls ``.cuda()`` on it
unconditionally; here it moves to an explicit device (default: the current HIP
device) without blocking.  ``gather_sum`` is the fused form of the only way the
encoder ever uses ``index_select_ND``: gather neighbour rows and sum them.
"""
from __future__ import annotations

import torch

from . import _capi as K
from . import graphprep
from .mpn import as_row_buffer, dense_view


def create_var(tensor, requires_grad=None, device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    out = tensor.to(device, non_blocking=True)
    if requires_grad is not None:
        out.requires_grad_(requires_grad)
    return out


def gather_sum(source: torch.Tensor, index) -> torch.Tensor:
    """``index_select_ND(source, 0, index).sum(dim=1)`` without the [M, K, H] intermediate.

    ``index`` is a zero-padded [M, K] neighbour table; row 0 of ``source`` must be the
    zero padding row, as everywhere in the reference's layout.  No autograd.
    """
    if source.device.type != "cuda":
        raise RuntimeError("gather_sum runs on an MI355X only; there is no CPU path")
    H = source.shape[1]
    ld = K.round_ld(H)
    rp, col = graphprep.padded_to_csr(graphprep._to_np(index))
    graphprep.check_range(col, source.shape[0], "index")
    csr = graphprep.upload_csr(rp, col, source.device)
    with torch.cuda.device(source.device), torch.no_grad():
        src = as_row_buffer(source, H, ld)
        out = K.rows(csr.n_rows, ld, source.device)
        K.csr_gather_sum(out.data_ptr(), src.data_ptr(), source.shape[0], None, csr.rowptr.data_ptr(),
                         csr.col.data_ptr(), csr.n_rows, H, ld)
    return dense_view(out, H)


class _CsrGatherSum(torch.autograd.Function):
    """out[i] = sum of source rows csr.col[csr.rowptr[i] : csr.rowptr[i+1]]; backward over csr_t."""

    @staticmethod
    def forward(ctx, source, csr, csr_t):
        H = source.shape[1]
        ld = K.round_ld(H)
        with torch.cuda.device(source.device):
            src = as_row_buffer(source, H, ld)
            out = K.rows(csr.n_rows, ld, source.device)
            K.csr_gather_sum(out.data_ptr(), src.data_ptr(), source.shape[0], None, csr.rowptr.data_ptr(),
                             csr.col.data_ptr(), csr.n_rows, H, ld, nnz=csr.nnz)
        ctx.csr_t, ctx.H, ctx.ld = csr_t, H, ld
        return dense_view(out, H)

    @staticmethod
    def backward(ctx, dout):
        csr_t, H, ld = ctx.csr_t, ctx.H, ctx.ld
        with torch.cuda.device(dout.device):
            d = as_row_buffer(dout, H, ld)
            dsrc = K.rows(csr_t.n_rows, ld, dout.device)
            K.csr_gather_sum(dsrc.data_ptr(), d.data_ptr(), d.shape[0], None, csr_t.rowptr.data_ptr(),
                             csr_t.col.data_ptr(), csr_t.n_rows, H, ld, nnz=csr_t.nnz)
        return dense_view(dsrc, H), None, None


def csr_gather_sum(source: torch.Tensor, csr, csr_t) -> torch.Tensor:
    """Differentiable neighbour sum over a prepared adjacency (``graphprep.DeviceCSR`` and its transpose)."""
    if source.device.type != "cuda":
        raise RuntimeError("csr_gather_sum runs on an MI355X only; there is no CPU path")
    return _CsrGatherSum.apply(source, csr, csr_t)


def index_select_ND(source, dim, index):
    """The reference's N-D gather (fast_jtnn/nnutils.py:13-18), kept for callers outside the
    encoder path (it materialises [M, K, H]; the encoder modules never call it)."""
    final_size = index.size() + source.size()[1:]
    return source.index_select(dim, index.view(-1)).view(final_size)
