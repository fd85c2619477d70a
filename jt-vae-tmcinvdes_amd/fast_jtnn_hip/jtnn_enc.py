"""Junction-tree encoder (tree message passing GRU) on the HIP kernels.

Drop-in for ``fast_jtnn.jtnn_enc.JTNNEncoder`` / ``GraphGRU``
(fast_jtnn/jtnn_enc.py:8-128): same constructors, same parameter names
(``embedding.weight``, ``outputNN.0.{weight,bias}``, ``GRU.W_z.{weight,bias}``,
``GRU.W_r.weight``, ``GRU.U_r.{weight,bias}``, ``GRU.W_h.{weight,bias}``), same
``forward(fnode, fmess, node_graph, mess_graph, scope) -> (tree_vecs, messages)``.

What is computed differently from the reference's op sequence (same mathematics):

* x = E[word of the message's source node] never changes over the 20 iterations, so
  ``W_z[:, :H] x + b_z``, ``W_r x``, ``W_h[:, :H] x + b_h`` and ``outputNN[:, :H] x + b``
  are computed once per *distinct word of the batch* (a few hundred rows) and looked
  up by the kernels; the reference recomputes ``W_r x`` on every message in every
  iteration (jtnn_enc.py:117).
* ``U_r`` is applied to each message once (a [NM,H]x[H,H] GEMM) and gathered, instead
  of to every (message, neighbour) pair (jtnn_enc.py:118: [NM*K, H] x [H, H]).
* gather, reset gate, both neighbour sums are one kernel; the update gate and the
  blend are epilogues of the two remaining GEMMs.
* only the root node of each tree needs ``outputNN`` (jtnn_enc.py:36-41 keeps
  ``node_vecs[st]`` only), so the output stage runs on B rows.
* a message is computed once, in the iteration in which all messages it reads have settled:
  the 20 synchronous iterations of the reference recompute every message until then, and
  nothing downstream ever reads those provisional values (``graphprep.build_tree_schedule``).
* backward is explicit; the 7 row buffers it needs per message are kept in HBM instead of
  autograd's [NM, K, H] saves per iteration.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _capi as K
from . import graphprep
from .mpn import as_row_buffer, dense_view, pack_weights


# "once": every message computed once, in the iteration in which it settles (falls back to "level" for batches
# with a tree deeper than the iteration count); "level": update the messages that can still change;
# "dense": the reference's schedule, all messages in all iterations
SCHEDULE = "once"


class _TreeGRU:
    """Forward/backward body of GraphGRU on padded row buffers, driven by the row lists of a schedule.

    Tables Pz, Pr, Ph are [n_words, ld]; wid maps a message to its table row.  Saved state (training)
    is one [NM, ld] buffer per quantity under the compute-once schedule, slabs [T(+1), NM, ld] of which
    only the rows of the schedule are written under the other two.
    """

    @staticmethod
    def forward(sched, NM, H, ld, dev, Pz, Pr, Ph, wid, mcsr, Wzh, Ur, bu, Whh, train, pad_row=0, issue=None):
        """``issue(args)``: with the one-call schedule, called with the sweep's argument block INSTEAD of launching
        iteration 0 and the sweep (the caller issues both inside a larger host call); see ``one_call``."""
        T = sched.T
        wid_p = wid.data_ptr()
        if T == 0:
            return K.rows(NM, ld, dev, zero=True), {}
        one_call = issue is not None and _TreeGRU.one_call(sched)
        # With the compute-once schedule a message has ONE saved state, that of the iteration in which it
        # settled: all "slabs" alias one [NM, ld] buffer (slab stride 0) and later iterations overwrite the
        # provisional rows iteration 0 left there.  The other schedules keep one slab per iteration.
        compact = sched.mode == "once"
        if train:
            n_slab = 1 if compact else T
            hs = torch.empty((1 if compact else T + 1, NM, ld), dtype=torch.float32, device=dev)
            zs, ps, ss, gs, ts, us = (torch.empty((n_slab, NM, ld), dtype=torch.float32, device=dev)
                                      for _ in range(6))
            stride = 0 if compact else NM * ld
            sf_p = sched.sf.data_ptr()
            st = dict(hs=hs, zs=zs, ps=ps, ss=ss, gs=gs, ts=ts, us=us, stride=stride)
            if not one_call:
                K.gru_iter0(Pz.data_ptr(), Ph.data_ptr(), wid_p, zs[0].data_ptr(), ps[0].data_ptr(),
                            hs[0 if compact else 1].data_ptr(), NM, H, ld, pad_row)
        else:
            # inference: one buffer each, updated in place (frozen rows simply keep their value)
            hs = torch.empty((1, NM, ld), dtype=torch.float32, device=dev)
            zs, ps, ss, gs, us = (torch.empty((1, NM, ld), dtype=torch.float32, device=dev) for _ in range(5))
            stride, sf_p, st = 0, None, {}
            if not one_call:
                K.gru_iter0(Pz.data_ptr(), Ph.data_ptr(), wid_p, None, None, hs[0].data_ptr(), NM, H, ld, pad_row)
        if compact and T > 1 and K.sweeps_enabled():
            # all levels in one host call: the same launches in the same order as the loop below
            lists = sched.__dict__.get("_fwd_lists")
            if lists is None:
                lists = sched.__dict__["_fwd_lists"] = (K.row_list_arrays(sched.F[1:T]), K.row_list_arrays(sched.U[1:T]))
            (f_rows, f_n), (u_rows, u_n) = lists
            a = K.TreeFwd(h=hs.data_ptr(), uh=us.data_ptr(), s=ss.data_ptr(), g=gs.data_ptr(),
                          t=ts.data_ptr() if train else None, z=zs.data_ptr(), p=ps.data_ptr(), sf=sf_p,
                          Pz=Pz.data_ptr(), Pr=Pr.data_ptr(), Ph=Ph.data_ptr(), wid=wid_p,
                          rowptr=mcsr.rowptr.data_ptr(), col=mcsr.col.data_ptr(), Wzh=Wzh.data_ptr(),
                          Whh=Whh.data_ptr(), Ur=Ur.data_ptr(), bu=None if bu is None else bu.data_ptr(), H=H, ld=ld,
                          pad_row=pad_row, n_levels=T - 1, f_rows=f_rows, f_n=f_n, u_rows=u_rows, u_n=u_n)
            if one_call:
                issue(a)
            else:
                K.tree_gru_forward(a)
            return hs[0], st
        for t in range(1, T):
            F, U = sched.F[t], sched.U[t]
            if F.n == 0:
                break  # every message has reached its fixed point
            slabbed = train and not compact
            i = t if slabbed else 0
            h_in = hs[t] if slabbed else hs[0]
            h_out = hs[t + 1] if slabbed else hs[0]
            # Uh = U_r h + b for the messages that changed in the last iteration and are read by someone
            K.gemm_nt(h_in.data_ptr(), ld, Ur.data_ptr(), ld, us[i].data_ptr(), ld, U.n, H, H,
                      bias=None if bu is None else bu.data_ptr(), a_rows=U.ptr(), c_rows=U.ptr())
            K.gru_gate_gather(hs.data_ptr(), us.data_ptr(), stride, t, sf_p, Pr.data_ptr(), wid_p,
                              mcsr.rowptr.data_ptr(), mcsr.col.data_ptr(), F.ptr(), F.n, ss[i].data_ptr(),
                              gs[i].data_ptr(), ts[i].data_ptr() if train else None, H, ld, nnz=F.nnz)
            # z = sigmoid(W_z[:, H:] s + Pz[wid]);  p = tanh(W_h[:, H:] g + Ph[wid]);  h' = (1-z) s + z p with the
            # padding row kept zero: both products and the blend in one launch
            K.gru_update(ss[i].data_ptr(), gs[i].data_ptr(), ld, Wzh.data_ptr(), Whh.data_ptr(), ld, Pz.data_ptr(),
                         Ph.data_ptr(), ld, wid_p, F.ptr(), F.n, zs[i].data_ptr(), ps[i].data_ptr(),
                         h_out.data_ptr(), H, pad_row)
        if train and not compact:
            h = K.rows(NM, ld, dev)
            K.select_slab_rows(h.data_ptr(), hs.data_ptr(), stride, sched.sel_final.data_ptr(), NM, H, ld)
        else:
            h = hs[0]
        return h, st

    @staticmethod
    def one_call(sched):
        """The forward sweep goes out as one host call (compute-once schedule, more than one iteration, launches not
        being timed one by one)."""
        return sched.mode == "once" and sched.T > 1 and K.sweeps_enabled()

    @staticmethod
    def backward(dh, sched, bs, NM, H, ld, dev, st, Pr, wid, mcsr_t, WzhT, WhhT, UrT, pad_row=0, issue=None):
        """dh: gradient w.r.t. the final h (row buffer indexed by message, modified in place).
        Returns compact (dz [nE], dp [nE], du [nJ]) and A_r [NM] for the weight / table gradients.
        ``issue(args, buffers)``: called with the sweep's argument block instead of issuing the sweep (one-call
        schedule only)."""
        T = sched.T
        hs, zs, ps, ss, gs, ts, us = (st[k] for k in ("hs", "zs", "ps", "ss", "gs", "ts", "us"))
        stride = st["stride"]
        sf_p = sched.sf.data_ptr()
        act_p = None if bs.active is None else bs.active.data_ptr()
        dz = K.rows(max(bs.nE, 1), ld, dev)
        dp = K.rows(max(bs.nE, 1), ld, dev)
        du = K.rows(max(bs.nJ, 1), ld, dev)
        a_r = K.rows(NM, ld, dev, zero=True)
        ds, dg, dhdir = K.rows(NM, ld, dev), K.rows(NM, ld, dev), K.rows(NM, ld, dev)
        wid_p = wid.data_ptr()
        if bs.once and K.sweeps_enabled():
            lists = bs.__dict__.get("_bwd_lists")
            if lists is None:
                n_lev = len(bs.E)
                lists = bs.__dict__["_bwd_lists"] = K.row_list_arrays(bs.E) + (
                    (K.C.c_int32 * n_lev)(*bs.n_dep), (K.C.c_int64 * n_lev)(*bs.offE), (K.C.c_int64 * n_lev)(*bs.offJ))
            e_rows, e_n, n_dep, off_e, off_j = lists
            a = K.TreeBwd(h=hs.data_ptr(), uh=us.data_ptr(), s=ss.data_ptr(), t=ts.data_ptr(), z=zs.data_ptr(),
                          p=ps.data_ptr(), sf=sf_p, active=act_p, Pr=Pr.data_ptr(), wid=wid_p,
                          trowptr=mcsr_t.rowptr.data_ptr(), tcol=mcsr_t.col.data_ptr(), WzhT=WzhT.data_ptr(),
                          WhhT=WhhT.data_ptr(), UrT=UrT.data_ptr(), dh=dh.data_ptr(), dz=dz.data_ptr(),
                          dp=dp.data_ptr(), du=du.data_ptr(), ds=ds.data_ptr(), dg=dg.data_ptr(),
                          dhdir=dhdir.data_ptr(), a_r=a_r.data_ptr(), T=T, H=H, ld=ld, pad_row=pad_row,
                          n_levels=len(bs.E), e_rows=e_rows, e_n=e_n, n_dep=n_dep, off_e=off_e, off_j=off_j)
            if issue is not None:  # the caller issues the sweep inside a larger host call
                issue(a, (dz, dp, du, a_r, ds, dg, dhdir))
                return dz, dp, du, a_r
            K.tree_gru_backward(a)
            return dz, dp, du, a_r
        if bs.once:
            # Every gradient-carrying message once, in reverse dependency order: when a level is reached all
            # messages that read its rows have been visited, so their gathered gradient is complete.  Saved
            # state has one row per message (slab stride 0); "iteration T" makes the kernels pick the state of
            # the iteration in which a message settled and treat messages without inputs as s = g = t = 0.
            for lev, E in enumerate(bs.E):
                if lev > 0:
                    du_t = K.ptr(du, bs.offJ[lev] * ld)
                    # dUh, and dh = (gradient from outside) + direct part, for rows whose readers are done
                    K.gru_bwd_gather(hs.data_ptr(), us.data_ptr(), stride, 0, sf_p, act_p, Pr.data_ptr(), wid_p,
                                     dg.data_ptr(), ds.data_ptr(), mcsr_t.rowptr.data_ptr(), mcsr_t.col.data_ptr(),
                                     E.ptr(), E.n, du_t, dhdir.data_ptr(), H, ld, nnz=bs.J[lev].nnz,
                                     dh_ext=dh.data_ptr())
                dz_t, dp_t = K.ptr(dz, bs.offE[lev] * ld), K.ptr(dp, bs.offE[lev] * ld)
                if lev > 0:
                    # dh of the level's rows, with the pointwise backward of blend / sigmoid / tanh in its epilogue
                    K.gru_bwd_dh(du_t, UrT.data_ptr(), ld, dhdir.data_ptr(), dh.data_ptr(), E.ptr(), E.n,
                                 zs.data_ptr(), ps.data_ptr(), ss.data_ptr(), stride, T, sf_p, dz_t, dp_t,
                                 ds.data_ptr(), H, ld, pad_row)
                else:
                    K.gru_bwd_pointwise(dh.data_ptr(), zs.data_ptr(), ps.data_ptr(), ss.data_ptr(), stride, T, sf_p,
                                        E.ptr(), E.n, dz_t, dp_t, ds.data_ptr(), H, ld, pad_row)
                nd = bs.n_dep[lev]  # leading rows of the level that read other messages
                if nd:
                    K.gemm_nt_pair(
                        K.nt_problem(dz_t, ld, WzhT.data_ptr(), ld, ds.data_ptr(), ld, nd, H, H, add=ds.data_ptr(),
                                     ld_add=ld, c_rows=E.ptr()),
                        K.nt_problem(dp_t, ld, WhhT.data_ptr(), ld, dg.data_ptr(), ld, nd, H, H, mode=K.EPI_MUL_ACC,
                                     aux0=ts.data_ptr(), out2=a_r.data_ptr(), ld_aux=ld, c_rows=E.ptr(), aux_sf=sf_p,
                                     aux_t=T, aux_slab_stride=stride))
            return dz, dp, du, a_r
        for t in range(T - 1, -1, -1):
            E = bs.E[t]
            if E.n == 0:
                break  # no gradient reaches earlier iterations
            dz_t, dp_t = K.ptr(dz, bs.offE[t] * ld), K.ptr(dp, bs.offE[t] * ld)
            K.gru_bwd_pointwise(dh.data_ptr(), zs.data_ptr(), ps.data_ptr(), ss.data_ptr(), stride, t, sf_p,
                                E.ptr(), E.n, dz_t, dp_t, ds.data_ptr(), H, ld, pad_row)
            if t == 0:
                break
            J = bs.J[t]
            # two independent products, issued together:
            #   ds += dzpre W_z[:, H:]
            #   dg = dppre W_h[:, H:];  A_r += dg * t   (gradient reaching W_r x through the reset gate)
            K.gemm_nt_pair(
                K.nt_problem(dz_t, ld, WzhT.data_ptr(), ld, ds.data_ptr(), ld, E.n, H, H, add=ds.data_ptr(), ld_add=ld,
                             c_rows=E.ptr()),
                K.nt_problem(dp_t, ld, WhhT.data_ptr(), ld, dg.data_ptr(), ld, E.n, H, H, mode=K.EPI_MUL_ACC,
                             aux0=ts.data_ptr(), out2=a_r.data_ptr(), ld_aux=ld, c_rows=E.ptr(), aux_sf=sf_p, aux_t=t,
                             aux_slab_stride=stride))
            if J.n == 0:
                break
            du_t = K.ptr(du, bs.offJ[t] * ld)
            K.gru_bwd_gather(hs.data_ptr(), us.data_ptr(), stride, t, sf_p, act_p, Pr.data_ptr(), wid_p,
                             dg.data_ptr(), ds.data_ptr(), mcsr_t.rowptr.data_ptr(), mcsr_t.col.data_ptr(),
                             J.ptr(), J.n, du_t, dhdir.data_ptr(), H, ld, nnz=J.nnz)
            # dh_t = dUh U_r + direct part, for the messages that were read
            K.gemm_nt(du_t, ld, UrT.data_ptr(), ld, dh.data_ptr(), ld, J.n, H, H, add=dhdir.data_ptr(), ld_add=ld,
                      c_rows=J.ptr())
        return dz, dp, du, a_r


def _keyed_sum(src, pieces, key_csr, n_keys, H, ld, dev):
    """out[k] = sum of src rows whose key is k (fixed order: pieces, then pieces of a key)."""
    part = K.rows(pieces.n_rows, ld, dev)
    K.csr_gather_sum(part.data_ptr(), src.data_ptr(), src.shape[0], None, pieces.rowptr.data_ptr(),
                     pieces.col.data_ptr(), pieces.n_rows, H, ld, nnz=pieces.nnz)
    out = K.rows(n_keys, ld, dev)
    K.csr_gather_sum(out.data_ptr(), part.data_ptr(), pieces.n_rows, None, key_csr.rowptr.data_ptr(),
                     key_csr.col.data_ptr(), n_keys, H, ld, nnz=key_csr.nnz)
    return out


CROWDED_TILES = os.environ.get("JTVAE_CROWDED_TILES", "1") == "1"
# From this many tree messages on (about 2 400 molecules) the graph encoder is still running when the FORWARD sweep
# reaches its short levels, and those take the crowded tiling as well (+1.7 % at 4096 molecules per step, +2.6 % in
# bulk encoding; in smaller batches the short levels run after the graph encoder has finished, and the one-tile
# kernels are faster there: -2 % at 512 to 2048 molecules).
CROWDED_SWEEP_MIN_MESSAGES = 100000


def crowded_tiles(crowded):
    """Row limit of the one-tile-per-block GEMM kernels for launches that run beside another stream's large
    products (``crowded``): 0, i.e. the 128 x 32 tiling, whose blocks fit on a CU that holds three blocks of a
    large product, while a one-tile block (115 KB of LDS) waits for a CU to drain - 100 to 250 us on the short
    levels of a backward sweep.  None (no change) otherwise.  The two tilings sum in different orders: results
    agree to rounding, not bit for bit."""
    return 0 if crowded else None


def _aside(fn, side_streams, dev, which=0):
    """fn() on side stream ``which`` behind everything the current stream holds so far; returns the event to wait
    for before its results are read (None: there is no such stream and fn ran in place)."""
    if len(side_streams) <= which:
        fn()
        return None
    s = side_streams[which]
    start = torch.cuda.Event()
    start.record(torch.cuda.current_stream(dev))
    s.wait_event(start)
    with torch.cuda.stream(s):
        fn()
        ev = torch.cuda.Event()
        ev.record(s)
    return ev


def _join(ev, dev):
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)


class _JTNNEncoderFunction(torch.autograd.Function):
    """forward(E, Wout, bout, Wz, bz, Wr, Ur, bu, Wh, bh) -> (tree_vecs [B,H], messages [NM,H])"""

    @staticmethod
    def forward(ctx, E, Wout, bout, Wz, bz, Wr, Ur_w, bu, Wh, bh, prep, depth, grad_on=True, side_streams=(),
                crowded=False):
        dev = E.device
        H = Wr.shape[0]
        ld = K.round_ld(H)
        P = prep
        NM, B, NW = P.n_mess, P.n_trees, P.n_words
        # grad mode is off inside forward(): the caller's mode comes in as an argument
        train = grad_on and any(ctx.needs_input_grad)
        ctx.set_materialize_grads(False)  # an unused output (messages in encode_latent) costs nothing
        if train:
            P.ensure_training()
        sched = graphprep.tree_schedule(P, depth, mode=SCHEDULE)

        # embedding rows of the words that occur in the batch            (jtnn_enc.py:27-28)
        E_p = as_row_buffer(E, H, ld)
        Eu = K.rows(NW, ld, dev)
        one_call = _TreeGRU.one_call(sched)
        if not one_call:
            K.csr_gather_sum(Eu.data_ptr(), E_p.data_ptr(), E.shape[0], None, P.word_csr.rowptr.data_ptr(),
                             P.word_csr.col.data_ptr(), NW, H, ld, nnz=NW)
        Wzx, Wzh, Whx, Whh, Wox, Woh, Wr_p, Ur_p = pack_weights(
            [(Wz, 0, H, False), (Wz, H, H, False), (Wh, 0, H, False), (Wh, H, H, False), (Wout, 0, H, False),
             (Wout, H, H, False), (Wr, 0, H, False), (Ur_w, 0, H, False)])
        Pz, Pr, Ph, Po = (K.rows(NW, ld, dev) for _ in range(4))
        if one_call:
            # the whole pass in one host call (jtvae_tree_encoder_forward): the launches of the branch below in the
            # same order, the helper stream forked and joined inside the call
            nsum, tv = K.rows(B, ld, dev), K.rows(B, ld, dev)
            side = side_streams[0].cuda_stream if len(side_streams) > 0 else None

            def issue(sweep):
                a = K.TreeEncFwd(NM=NM, B=B, NW=NW, V=E.shape[0], H=H, ld=ld,
                                 crowded=(3 if NM >= CROWDED_SWEEP_MIN_MESSAGES else 1) if crowded else 0,
                                 train=1 if train else 0, E=E_p.data_ptr(), word=K.csr_arg(P.word_csr, NW),
                                 root=K.csr_arg(P.rcsr, B), Wzx=Wzx.data_ptr(), Whx=Whx.data_ptr(), Wox=Wox.data_ptr(),
                                 Wr=Wr_p.data_ptr(), Woh=Woh.data_ptr(), bz=bz.detach().data_ptr(),
                                 bh=bh.detach().data_ptr(), bout=bout.detach().data_ptr(),
                                 wid_root=P.wid_root.data_ptr(), sweep=K.C.pointer(sweep), Eu=Eu.data_ptr(),
                                 Po=Po.data_ptr(), nsum=nsum.data_ptr(), tv=tv.data_ptr())
                K.tree_encoder_forward(a, side)

            h, st = _TreeGRU.forward(sched, NM, H, ld, dev, Pz, Pr, Ph, P.wid_m, P.mcsr, Wzh, Ur_p, bu.detach(), Whh,
                                     train, issue=issue)
        else:
            h, st, nsum, tv = _JTNNEncoderFunction._forward_by_launch(
                sched, P, NM, B, NW, H, ld, dev, Eu, (Pz, Pr, Ph, Po), (Wzx, Wzh, Whx, Whh, Wox, Woh, Wr_p, Ur_p),
                (bz, bh, bout, bu), train, side_streams, crowded)
        if train:
            ctx.prep, ctx.depth, ctx.H, ctx.ld = P, depth, H, ld
            ctx.st, ctx.sched = st, sched
            ctx.keep = (Eu, Pr, nsum, tv, Wout, Wz, Wr, Ur_w, Wh)
            ctx.vocab = E.shape[0]
            ctx.side_streams = tuple(side_streams)
            ctx.crowded = crowded
        return dense_view(tv, H), dense_view(h, H)

    @staticmethod
    def _forward_by_launch(sched, P, NM, B, NW, H, ld, dev, Eu, tables, weights, biases, train, side_streams, crowded):
        """The tables, the sweep and the output stage issued launch by launch (any schedule; the form that is timed)."""
        Pz, Pr, Ph, Po = tables
        Wzx, Wzh, Whx, Whh, Wox, Woh, Wr_p, Ur_p = weights
        bz, bh, bout, bu = biases
        # the reset-gate table is first read by level 1 of the sweep and the output table after it: with a side
        # stream they leave the chain of dependent launches that the sweep is
        # With the graph encoder on another stream, the table products start beside its first large products:
        # they then take the tiling that finds room on a busy CU (crowded_tiles).
        with K.small_tile_rows(crowded_tiles(crowded)):
            ev_r = _aside(lambda: K.gemm_nt(Eu.data_ptr(), ld, Wr_p.data_ptr(), ld, Pr.data_ptr(), ld, NW, H, H),
                          side_streams, dev)
            ev_o = _aside(lambda: K.gemm_nt(Eu.data_ptr(), ld, Wox.data_ptr(), ld, Po.data_ptr(), ld, NW, H, H,
                                            bias=bout.detach().data_ptr()), side_streams, dev)
            K.gemm_nt(Eu.data_ptr(), ld, Wzx.data_ptr(), ld, Pz.data_ptr(), ld, NW, H, H, bias=bz.detach().data_ptr())
            K.gemm_nt(Eu.data_ptr(), ld, Whx.data_ptr(), ld, Ph.data_ptr(), ld, NW, H, H, bias=bh.detach().data_ptr())
        _join(ev_r, dev)
        with K.small_tile_rows(crowded_tiles(crowded and NM >= CROWDED_SWEEP_MIN_MESSAGES)):
            h, st = _TreeGRU.forward(sched, NM, H, ld, dev, Pz, Pr, Ph, P.wid_m, P.mcsr, Wzh, Ur_p, bu.detach(), Whh,
                                     train)
        _join(ev_o, dev)

        # output stage on the root nodes only                             (jtnn_enc.py:31-41)
        nsum = K.rows(B, ld, dev)
        K.csr_gather_sum(nsum.data_ptr(), h.data_ptr(), NM, None, P.rcsr.rowptr.data_ptr(),
                         P.rcsr.col.data_ptr(), B, H, ld, nnz=P.rcsr.nnz)
        tv = K.rows(B, ld, dev)
        K.gemm_nt(nsum.data_ptr(), ld, Woh.data_ptr(), ld, tv.data_ptr(), ld, B, H, H, add=Po.data_ptr(),
                  ld_add=ld, add_idx=P.wid_root.data_ptr(), act=K.ACT_RELU)
        return h, st, nsum, tv

    @staticmethod
    def backward(ctx, dtv, dmess):
        P, depth, H, ld = ctx.prep, ctx.depth, ctx.H, ctx.ld
        Eu, Pr, nsum, tv, Wout, Wz, Wr, Ur_w, Wh = ctx.keep
        st, sched = ctx.st, ctx.sched
        dev = Eu.device
        NM, B, NW, V = P.n_mess, P.n_trees, P.n_words, ctx.vocab
        T = depth
        # which messages carry gradient: all of them if the caller used `messages`, else those into a root
        bs = graphprep.backward_schedule(P, sched, all_messages=dmess is not None)

        dWout = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
        dWz = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
        dWh = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
        dWr = torch.empty((H, H), dtype=torch.float32, device=dev)
        dUr = torch.empty((H, H), dtype=torch.float32, device=dev)
        dbout, dbz, dbu, dbh = (torch.empty((H,), dtype=torch.float32, device=dev) for _ in range(4))

        # transposed weights of every dgrad GEMM of this backward pass
        WohT, WzhT, WhhT, UrT, WzxT, WhxT, WoxT, WrT = pack_weights(
            [(Wout, H, H, True), (Wz, H, H, True), (Wh, H, H, True), (Ur_w, 0, H, True), (Wz, 0, H, True),
             (Wh, 0, H, True), (Wout, 0, H, True), (Wr, 0, H, True)])
        if K.sweeps_enabled() and T >= 1 and bs.once and bs.nE > 0 and bs.n_z > 0 and bs.n_u > 0:
            return _JTNNEncoderFunction._backward_one_call(
                ctx, dtv, dmess, bs, (dWout, dWz, dWh, dWr, dUr, dbout, dbz, dbu, dbh),
                (WohT, WzhT, WhhT, UrT, WzxT, WhxT, WoxT, WrT))
        # root stage: relu mask, W_out[:, H:] gradient, back to the messages that feed a root
        dpre_root = K.rows(B, ld, dev)
        if dtv is None:
            dpre_root.zero_()
        else:
            dtv_p = as_row_buffer(dtv, H, ld)
            K.csr_gather_sum(dpre_root.data_ptr(), dtv_p.data_ptr(), B, None, P.ident.rowptr.data_ptr(),
                             P.ident.col.data_ptr(), B, H, ld, relu_mask=tv.data_ptr(), nnz=B)
        tables = {}
        home = torch.cuda.current_stream(dev)

        def table_grad(name, src, pieces, key_csr, dW, lddw, db):
            # gradient of one per-word table, and of the weight block (and bias) that produced it
            dPt = _keyed_sum(src, pieces, key_csr, NW, H, ld, dev) if src is not None else K.rows(NW, ld, dev, zero=True)
            K.gemm_tn(dPt.data_ptr(), ld, Eu.data_ptr(), ld, dW.data_ptr(), lddw, NW, H, H, False, dev,
                      dbias=None if db is None else db.data_ptr())
            if torch.cuda.current_stream(dev) != home:
                dPt.record_stream(home)  # made on a side stream, read on this pass's own stream below
            tables[name] = dPt

        def root_stage_grads():
            # W_out[:, H:] and everything that hangs on the output table: nothing in the sweep waits for them
            K.gemm_tn(dpre_root.data_ptr(), ld, nsum.data_ptr(), ld, K.ptr(dWout, H), 2 * H, B, H, H, False, dev)
            table_grad("o", dpre_root, P.key_r_pieces, P.key_r, dWout, 2 * H, dbout)

        # With a side stream, work that the chain of the sweep does not wait for runs beside it: the root stage's
        # weight and table gradients now, the recurrence's weight gradients after the sweep.  The same launches
        # on the same data either way.
        ev_root = _aside(root_stage_grads, ctx.side_streams, dev)
        dnsum = K.rows(B, ld, dev)
        K.gemm_nt(dpre_root.data_ptr(), ld, WohT.data_ptr(), ld, dnsum.data_ptr(), ld, B, H, H)
        dh = K.rows(NM, ld, dev)
        if dmess is not None:
            dm = dmess if dmess.stride(1) == 1 and dmess.stride(0) >= H else dmess.contiguous()
            K.copy_rows(dm.data_ptr(), dm.stride(0), dh.data_ptr(), ld, NM, H)
        K.csr_gather_sum(dh.data_ptr(), dnsum.data_ptr(), B, None, P.rcsr_t.rowptr.data_ptr(),
                         P.rcsr_t.col.data_ptr(), NM, H, ld, accumulate=dmess is not None, nnz=P.rcsr_t.nnz)

        if T >= 1:
            # the backward sweep runs beside the graph encoder's backward pass from its first level to its last
            with K.small_tile_rows(crowded_tiles(ctx.crowded)):
                dz, dp, du, a_r = _TreeGRU.backward(dh, sched, bs, NM, H, ld, dev, st, Pr, P.wid_m, P.mcsr_t, WzhT,
                                                    WhhT, UrT)

        def gru_weight_grads():
            # one contraction each over every (iteration, message) pair that carried gradient, operands picked
            # through row lists
            if T >= 1 and bs.n_z > 0:
                K.gemm_tn(dz.data_ptr(), ld, st["ss"].data_ptr(), ld, K.ptr(dWz, H), 2 * H, bs.n_z, H, H, False, dev,
                          x_rows=bs.zx.data_ptr(), y_rows=bs.zy.data_ptr())
                K.gemm_tn(dp.data_ptr(), ld, st["gs"].data_ptr(), ld, K.ptr(dWh, H), 2 * H, bs.n_z, H, H, False, dev,
                          x_rows=bs.zx.data_ptr(), y_rows=bs.zy.data_ptr())
            else:
                dWz[:, H:].zero_()
                dWh[:, H:].zero_()
            if T >= 1 and bs.n_u > 0:
                # the row pairs visit every compact du row exactly once, so the fused column sum is U_r's bias
                # gradient
                K.gemm_tn(du.data_ptr(), ld, st["hs"].data_ptr(), ld, dUr.data_ptr(), H, bs.n_u, H, H, False, dev,
                          x_rows=bs.ux.data_ptr(), y_rows=bs.uy.data_ptr(), dbias=dbu.data_ptr())
            else:
                dUr.zero_()
                dbu.zero_()

        # after the sweep: a run of short launches (three weight-gradient contractions, and per table two keyed
        # sums and one more contraction) that would leave most of the GPU idle one after another
        ev_w = _aside(gru_weight_grads, ctx.side_streams, dev)
        swept = T >= 1 and bs.nE > 0
        table_grad("z", dz if swept else None, bs.key_pieces, bs.key, dWz, 2 * H, dbz)
        table_grad("h", dp if swept else None, bs.key_pieces, bs.key, dWh, 2 * H, dbh)
        table_grad("r", a_r if swept else None, P.key_m_pieces, P.key_m, dWr, H, None)
        _join(ev_root, dev)
        dPz, dPh, dPo, dPr = (tables[k] for k in "zhor")
        # embedding gradient: dEu = sum_tables dP W[:, :H], scattered to the rows of the words used
        dEu = K.rows(NW, ld, dev)
        first = True
        for dPt, WT in ((dPz, WzxT), (dPh, WhxT), (dPo, WoxT), (dPr, WrT)):
            K.gemm_nt(dPt.data_ptr(), ld, WT.data_ptr(), ld, dEu.data_ptr(), ld, NW, H, H,
                      add=None if first else dEu.data_ptr(), ld_add=ld)
            first = False
        dE_p = K.rows(V, ld, dev)
        K.csr_gather_sum(dE_p.data_ptr(), dEu.data_ptr(), NW, None, P.word_csr_t.rowptr.data_ptr(),
                         P.word_csr_t.col.data_ptr(), V, H, ld, nnz=NW)
        dE = torch.empty((V, H), dtype=torch.float32, device=dev)
        K.copy_rows(dE_p.data_ptr(), ld, dE.data_ptr(), H, V, H)
        # the recurrence's weight gradients are not read here: their stream is joined last, so the embedding chain
        # above runs beside them
        _join(ev_w, dev)
        return dE, dWout, dbout, dWz, dbz, dWr, dUr, dbu, dWh, dbh, None, None, None, None, None


    @staticmethod
    def _backward_one_call(ctx, dtv, dmess, bs, grads, packed):
        """The backward pass as ONE host call (jtvae_tree_encoder_backward): the launches of ``backward`` in the same
        order on the same data, the helper stream forked and joined inside the call."""
        P, depth, H, ld = ctx.prep, ctx.depth, ctx.H, ctx.ld
        Eu, Pr, nsum, tv, _, _, _, _, _ = ctx.keep
        st, sched = ctx.st, ctx.sched
        dev = Eu.device
        NM, B, NW, V = P.n_mess, P.n_trees, P.n_words, ctx.vocab
        dWout, dWz, dWh, dWr, dUr, dbout, dbz, dbu, dbh = grads
        WohT, WzhT, WhhT, UrT, WzxT, WhxT, WoxT, WrT = packed
        dh = K.rows(NM, ld, dev)
        if dmess is not None and not (dmess.stride(1) == 1 and dmess.stride(0) >= H):
            dmess = dmess.contiguous()  # e.g. a broadcast gradient (stride 0)
        n_side = P.key_r_pieces.n_rows
        n_main = max(bs.key_pieces.n_rows, P.key_m_pieces.n_rows)
        # every scratch row buffer of the pass in one allocation
        sizes = (B, B, n_side, n_main, NW, NW, NW, NW, NW, V)
        arena = K.rows(sum(max(n, 1) for n in sizes), ld, dev)
        ptrs, row = [], 0
        for n in sizes:
            ptrs.append(arena.data_ptr() + row * ld * 4)
            row += max(n, 1)
        dpre_root, dnsum, part_side, part_main, dPo, dPz, dPh, dPr, dEu, dE_p = ptrs
        shapes = ((B, H, H), (NW, H, H), (bs.n_z, H, H), (bs.n_z, H, H), (bs.n_u, H, H), (NW, H, H), (NW, H, H),
                  (NW, H, H))
        ws_bytes = [max(K.gemm_tn_workspace_bytes(*sh), 16) for sh in shapes]
        offs, tot = [], 0
        for nb in ws_bytes:
            offs.append(tot)
            tot += (nb + 255) // 256 * 256
        ws = torch.empty((tot,), dtype=torch.uint8, device=dev)
        dE = torch.empty((V, H), dtype=torch.float32, device=dev)
        dtv_p = None if dtv is None else as_row_buffer(dtv, H, ld)
        side = ctx.side_streams[0].cuda_stream if len(ctx.side_streams) > 0 else None
        NWk = lambda c: K.csr_arg(c, NW)  # noqa: E731
        out = {}

        def issue(sweep, buffers):
            a = K.TreeEncBwd(
                NM=NM, B=B, NW=NW, V=V, H=H, ld=ld, crowded=1 if ctx.crowded else 0, Eu=Eu.data_ptr(),
                nsum=nsum.data_ptr(), tv=tv.data_ptr(), dtv=None if dtv_p is None else dtv_p.data_ptr(),
                dmess=None if dmess is None else dmess.data_ptr(), ld_dmess=0 if dmess is None else dmess.stride(0),
                WohT=WohT.data_ptr(), WzxT=WzxT.data_ptr(), WhxT=WhxT.data_ptr(), WoxT=WoxT.data_ptr(),
                WrT=WrT.data_ptr(), ident=K.csr_arg(P.ident, B), root_t=K.csr_arg(P.rcsr_t, NM),
                word_t=K.csr_arg(P.word_csr_t, V), o_pieces=K.csr_arg(P.key_r_pieces), o_key=NWk(P.key_r),
                zh_pieces=K.csr_arg(bs.key_pieces), zh_key=NWk(bs.key), r_pieces=K.csr_arg(P.key_m_pieces),
                r_key=NWk(P.key_m), sweep=K.C.pointer(sweep), n_e=max(bs.nE, 1), ss=st["ss"].data_ptr(),
                gs=st["gs"].data_ptr(), hs=st["hs"].data_ptr(), zx=bs.zx.data_ptr(), zy=bs.zy.data_ptr(),
                ux=bs.ux.data_ptr(), uy=bs.uy.data_ptr(), n_z=bs.n_z, n_u=bs.n_u, dpre_root=dpre_root, dnsum=dnsum,
                part_side=part_side, part_main=part_main, dPo=dPo, dPz=dPz, dPh=dPh, dPr=dPr, dEu=dEu, dE_p=dE_p,
                ws=(K.C.c_void_p * 8)(*[ws.data_ptr() + o for o in offs]), ws_bytes=(K.C.c_size_t * 8)(*ws_bytes),
                dWout=dWout.data_ptr(), dWz=dWz.data_ptr(), dWh=dWh.data_ptr(), dWr=dWr.data_ptr(),
                dUr=dUr.data_ptr(), dbout=dbout.data_ptr(), dbz=dbz.data_ptr(), dbh=dbh.data_ptr(),
                dbu=dbu.data_ptr(), dE=dE.data_ptr())
            K.tree_encoder_backward(a, side)
            out["keep"] = buffers  # alive until the call has been issued

        _TreeGRU.backward(dh, sched, bs, NM, H, ld, dev, st, Pr, P.wid_m, P.mcsr_t, WzhT, WhhT, UrT, issue=issue)
        # Scratch that the helper stream read or wrote goes back to the allocator of the stream it came from: the call
        # has joined the helper into this stream, so whatever reuses the memory is ordered behind the helper's work.
        return dE, dWout, dbout, dWz, dbz, dWr, dUr, dbu, dWh, dbh, None, None, None, None, None


class _MessageGRUFunction(torch.autograd.Function):
    """forward(E, Wz, bz, Wr, br | None, Ur, bu | None, Wh, bh) -> messages [NM, H]

    The GRU of nnutils.GRU / GraphGRU over the messages of a dependency graph without cycles
    (``graphprep.DagPrep``), each message computed once, level by level.  It serves the teacher-forced
    tree decoder (fast_jtnn/jtnn_dec.py:76-147 with fast_jtnn/nnutils.py:56-71), whose reset gate has
    its bias on W_r and none on U_r - the other way round from GraphGRU - hence both are optional.
    """

    @staticmethod
    def forward(ctx, E, Wz, bz, Wr, br, Ur_w, bu, Wh, bh, prep, grad_on=True, depth=None):
        """depth: number of GRU iterations (None: as many as the longest dependency chain, so that every message
        settles; a smaller number follows the truncated recurrence iteration by iteration)."""
        dev = E.device
        H = Wr.shape[0]
        ld = K.round_ld(H)
        P = prep
        NM, NW = P.n_mess, P.n_words
        train = grad_on and any(ctx.needs_input_grad)
        sched = graphprep.tree_schedule(P, P.depth if depth is None else depth, mode=SCHEDULE if depth is not None else "once")
        E_p = as_row_buffer(E, H, ld)
        Eu = K.rows(NW, ld, dev)
        K.csr_gather_sum(Eu.data_ptr(), E_p.data_ptr(), E.shape[0], None, P.word_csr.rowptr.data_ptr(),
                         P.word_csr.col.data_ptr(), NW, H, ld, nnz=NW)
        Wzx, Wzh, Whx, Whh, Wr_p, Ur_p = pack_weights(
            [(Wz, 0, H, False), (Wz, H, H, False), (Wh, 0, H, False), (Wh, H, H, False), (Wr, 0, H, False),
             (Ur_w, 0, H, False)])
        Pz, Pr, Ph = (K.rows(NW, ld, dev) for _ in range(3))
        K.gemm_nt(Eu.data_ptr(), ld, Wzx.data_ptr(), ld, Pz.data_ptr(), ld, NW, H, H, bias=bz.detach().data_ptr())
        K.gemm_nt(Eu.data_ptr(), ld, Wr_p.data_ptr(), ld, Pr.data_ptr(), ld, NW, H, H,
                  bias=None if br is None else br.detach().data_ptr())
        K.gemm_nt(Eu.data_ptr(), ld, Whx.data_ptr(), ld, Ph.data_ptr(), ld, NW, H, H, bias=bh.detach().data_ptr())
        h, st = _TreeGRU.forward(sched, NM, H, ld, dev, Pz, Pr, Ph, P.wid_m, P.mcsr, Wzh, Ur_p,
                                 None if bu is None else bu.detach(), Whh, train)
        if train:
            ctx.prep, ctx.H, ctx.ld, ctx.st, ctx.sched = P, H, ld, st, sched
            ctx.keep = (Eu, Pr, Wz, Wr, Ur_w, Wh)
            ctx.vocab, ctx.has_br, ctx.has_bu = E.shape[0], br is not None, bu is not None
        return dense_view(h, H)

    @staticmethod
    def backward(ctx, dmess):
        P, H, ld, st, sched = ctx.prep, ctx.H, ctx.ld, ctx.st, ctx.sched
        Eu, Pr, Wz, Wr, Ur_w, Wh = ctx.keep
        dev = Eu.device
        NM, NW, V = P.n_mess, P.n_words, ctx.vocab
        bs = graphprep.backward_schedule(P, sched, all_messages=True)
        dWz = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
        dWh = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
        dWr = torch.empty((H, H), dtype=torch.float32, device=dev)
        dUr = torch.empty((H, H), dtype=torch.float32, device=dev)
        dbz, dbh, dbr, dbu = (torch.empty((H,), dtype=torch.float32, device=dev) for _ in range(4))
        WzhT, WhhT, UrT, WzxT, WhxT, WrT = pack_weights(
            [(Wz, H, H, True), (Wh, H, H, True), (Ur_w, 0, H, True), (Wz, 0, H, True), (Wh, 0, H, True),
             (Wr, 0, H, True)])
        dh = K.rows(NM, ld, dev)  # private copy: the backward sweep accumulates into it
        dm = dmess if dmess.dim() == 2 and dmess.stride(1) == 1 and dmess.stride(0) >= H else dmess.contiguous()
        K.copy_rows(dm.data_ptr(), dm.stride(0), dh.data_ptr(), ld, NM, H)
        dz, dp, du, a_r = _TreeGRU.backward(dh, sched, bs, NM, H, ld, dev, st, Pr, P.wid_m, P.mcsr_t, WzhT, WhhT,
                                            UrT)
        if bs.n_z > 0:
            K.gemm_tn(dz.data_ptr(), ld, st["ss"].data_ptr(), ld, K.ptr(dWz, H), 2 * H, bs.n_z, H, H, False, dev,
                      x_rows=bs.zx.data_ptr(), y_rows=bs.zy.data_ptr())
            K.gemm_tn(dp.data_ptr(), ld, st["gs"].data_ptr(), ld, K.ptr(dWh, H), 2 * H, bs.n_z, H, H, False, dev,
                      x_rows=bs.zx.data_ptr(), y_rows=bs.zy.data_ptr())
        else:
            dWz[:, H:].zero_()
            dWh[:, H:].zero_()
        if bs.n_u > 0:
            K.gemm_tn(du.data_ptr(), ld, st["hs"].data_ptr(), ld, dUr.data_ptr(), H, bs.n_u, H, H, False, dev,
                      x_rows=bs.ux.data_ptr(), y_rows=bs.uy.data_ptr(), dbias=dbu.data_ptr())
        else:
            dUr.zero_()
            dbu.zero_()
        dPz = _keyed_sum(dz, bs.key_pieces, bs.key, NW, H, ld, dev)
        dPh = _keyed_sum(dp, bs.key_pieces, bs.key, NW, H, ld, dev)
        dPr = _keyed_sum(a_r, P.key_m_pieces, P.key_m, NW, H, ld, dev)
        for dPt, dW, ldw, db in ((dPz, dWz, 2 * H, dbz), (dPh, dWh, 2 * H, dbh), (dPr, dWr, H, dbr)):
            K.gemm_tn(dPt.data_ptr(), ld, Eu.data_ptr(), ld, dW.data_ptr(), ldw, NW, H, H, False, dev,
                      dbias=db.data_ptr())
        dEu = K.rows(NW, ld, dev)
        for i, (dPt, WT) in enumerate(((dPz, WzxT), (dPh, WhxT), (dPr, WrT))):
            K.gemm_nt(dPt.data_ptr(), ld, WT.data_ptr(), ld, dEu.data_ptr(), ld, NW, H, H,
                      add=dEu.data_ptr() if i else None, ld_add=ld)
        dE_p = K.rows(V, ld, dev)
        K.csr_gather_sum(dE_p.data_ptr(), dEu.data_ptr(), NW, None, P.word_csr_t.rowptr.data_ptr(),
                         P.word_csr_t.col.data_ptr(), V, H, ld, nnz=NW)
        dE = torch.empty((V, H), dtype=torch.float32, device=dev)
        K.copy_rows(dE_p.data_ptr(), ld, dE.data_ptr(), H, V, H)
        return (dE, dWz, dbz, dWr, dbr if ctx.has_br else None, dUr, dbu if ctx.has_bu else None, dWh, dbh, None,
                None, None)


class GraphGRU(nn.Module):
    """Parameter container with the reference's layout (fast_jtnn/jtnn_enc.py:95-105).

    ``forward(h, x, mess_graph)`` keeps the reference signature for stand-alone use:
    ``x`` are per-message input rows, ``h`` must be the zero start state.
    """

    def __init__(self, input_size, hidden_size, depth):
        super(GraphGRU, self).__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.depth = depth
        self.W_z = nn.Linear(input_size + hidden_size, hidden_size)
        self.W_r = nn.Linear(input_size, hidden_size, bias=False)
        self.U_r = nn.Linear(hidden_size, hidden_size)
        self.W_h = nn.Linear(input_size + hidden_size, hidden_size)

    def forward(self, h, x, mess_graph):
        """Stand-alone use with explicit per-message inputs (fast_jtnn/jtnn_enc.py:107-128): every message is its
        own table row, gradients reach ``x`` and the weights.  ``h`` must be the zero start state."""
        if self.input_size != self.hidden_size:
            raise ValueError("GraphGRU kernels need input_size == hidden_size")
        dev = self.W_r.weight.device
        if dev.type != "cuda":
            raise RuntimeError("fast_jtnn_hip.GraphGRU runs on an MI355X only; there is no CPU path")
        H = self.hidden_size
        mg = graphprep._to_np(mess_graph)
        NM = mg.shape[0]
        if tuple(h.shape) != (NM, H) or tuple(x.shape) != (NM, H):
            raise ValueError("h and x must be [%d, %d]" % (NM, H))
        if bool((h != 0).any()):
            raise ValueError("GraphGRU kernels start from the zero state (as JTNNEncoder does, "
                             "fast_jtnn/jtnn_enc.py:25): h must be all zeros")
        if self.depth < 1:
            return h.to(dev) * 1.0
        import numpy as np

        def build():
            rp, col = graphprep.padded_to_csr(mg, NM)
            return graphprep.prepare_message_dag(np.arange(NM, dtype=np.int64), rp, col, NM, dev)

        key = mess_graph if isinstance(mess_graph, torch.Tensor) else None
        prep = build() if key is None else graphprep.cached_prep(key, ("graph_gru", str(dev)), build)
        with torch.cuda.device(dev):
            return _MessageGRUFunction.apply(x.to(dev), self.W_z.weight, self.W_z.bias, self.W_r.weight, None,
                                             self.U_r.weight, self.U_r.bias, self.W_h.weight, self.W_h.bias, prep,
                                             torch.is_grad_enabled(), self.depth)


class JTNNEncoder(nn.Module):
    def __init__(self, hidden_size, depth, embedding):
        super(JTNNEncoder, self).__init__()
        self.hidden_size = hidden_size
        self.depth = depth
        self.embedding = embedding
        self.outputNN = nn.Sequential(nn.Linear(2 * hidden_size, hidden_size), nn.ReLU())
        self.GRU = GraphGRU(hidden_size, hidden_size, depth=depth)

    # Streams the backward pass may use beside its own for the independent groups of short launches that follow
    # its level sweep (set by a caller that overlaps the two encoders; empty: everything on one stream).
    side_streams = ()
    # Whether the short products take the tiling that finds room on a CU busy with another stream's large products
    # (crowded_tiles).  None: exactly when side streams are set, unless JTVAE_CROWDED_TILES=0; True / False: always /
    # never, whatever the streams - the setting under which a one-stream and a two-stream pass issue the same
    # launches and so return the same bits.
    crowded = None

    def forward(self, fnode, fmess, node_graph, mess_graph, scope):
        dev = self.GRU.W_r.weight.device
        if dev.type != "cuda":
            raise RuntimeError("fast_jtnn_hip.JTNNEncoder runs on an MI355X only (module is on %s); there is "
                               "no CPU path" % dev)
        holder = (fnode, fmess, node_graph, mess_graph, scope)
        if SCHEDULE == "once":
            # trees deeper than the iteration count take the iteration-by-iteration schedule on their own
            key = mess_graph if isinstance(mess_graph, torch.Tensor) else None
            build = lambda: graphprep.split_deep_trees(holder, self.depth)  # noqa: E731
            split = build() if key is None else graphprep.cached_prep(key, ("deep_split", self.depth), build)
            if split is not None:
                return self._forward_split(split, len(scope), int(mess_graph.shape[0]), dev)
        return self._forward_holder(holder, dev)

    def _forward_split(self, split, n_trees, n_mess, dev):
        H = self.hidden_size
        tree_vecs = torch.zeros((n_trees, H), dtype=torch.float32, device=dev)
        messages = torch.zeros((n_mess, H), dtype=torch.float32, device=dev)  # row 0: the padding message
        for sub, tree_ids, mess_ids in split:
            tv, ms = self._forward_holder(sub, dev)
            tree_vecs = tree_vecs.index_copy(0, tree_ids.to(dev), tv)
            messages = messages.index_copy(0, mess_ids.to(dev), ms[1:])
        return tree_vecs, messages

    def _forward_holder(self, holder, dev):
        fnode, fmess, node_graph, mess_graph, scope = holder
        V = self.embedding.weight.shape[0]
        key = mess_graph if isinstance(mess_graph, torch.Tensor) else None
        build = lambda: graphprep.prepare_tree((fnode, fmess, node_graph, mess_graph, scope), dev, V,  # noqa: E731
                                               torch.is_grad_enabled())
        prep = build() if key is None else graphprep.cached_prep(key, ("tree", str(dev), V), build)
        g = self.GRU
        lin = self.outputNN[0]
        with torch.cuda.device(dev):
            return _JTNNEncoderFunction.apply(self.embedding.weight, lin.weight, lin.bias, g.W_z.weight,
                                              g.W_z.bias, g.W_r.weight, g.U_r.weight, g.U_r.bias,
                                              g.W_h.weight, g.W_h.bias, prep, self.depth, torch.is_grad_enabled(),
                                              self.side_streams, self._crowded())

    def _crowded(self):
        if self.crowded is None:
            return CROWDED_TILES and len(self.side_streams) > 0
        return bool(self.crowded)

    @staticmethod
    def tensorize(tree_batch):
        from .tensorize import jtenc_tensorize

        return jtenc_tensorize(tree_batch)

    @staticmethod
    def tensorize_nodes(node_batch, scope):
        from .tensorize import jtenc_tensorize_nodes

        return jtenc_tensorize_nodes(node_batch, scope)
