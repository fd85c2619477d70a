"""From flat graph storage straight to what the kernels read: holders whose tables never exist.

``datautils.tensorize`` of the reference (fast_jtnn/datautils.py:140-163) builds padded int64 tables on the
host and every ``forward`` copies them to the GPU (``create_var``, fast_jtnn/nnutils.py:6-10).  The modules of
this package turn such tables into int32 CSR first (``graphprep``).  When a batch comes from flat storage
(``flat.FlatShard`` + molecule positions) the detour through the tables is skipped: the native builders write
byte feature rows and CSR arrays directly into pinned staging memory, one copy takes them to the device, and
the holders handed to the model are *shape-only*: tuples in the reference's holder layout whose tensors own no
storage (``[n, 0]``) and carry the prepared device form under the keys the modules look it up by.  ``model(
*holder)`` therefore works unchanged, and a module asked for a device or vocabulary the holder was not prepared
for fails loudly (there are no tables to fall back to).

    shard = flat.shard_from_trees(trees, with_cands=True)          # or arrays loaded from disk
    x_batch = flatprep.training_batch(model, shard, ids)            # JTNNVAE.forward(x_batch, beta)
    jt, mp = flatprep.encoder_batch(model, shard, ids, train=False) # model.encode(jt, mp)
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi as K
from . import flat, graphprep
from .graphprep import DeviceCSR, GraphMPNPrep

_TORCH_OF = {np.dtype(np.uint8): torch.uint8, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32,
             np.dtype(np.int64): torch.int64}


class _Stage:
    """``alloc`` for the native builders: every array is a block of the current packed upload (pinned staging +
    its device image), so results are written where the one host-to-device copy picks them up.  On a CPU device
    (tests without a GPU) the arrays are plain numpy and the "device" views alias them."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.up = graphprep._TLS.upload
        self.dev = {}

    def alloc(self, name, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        if self.up is None:
            h = np.empty(shape, dtype)
            self.dev[name] = torch.from_numpy(h)
            return h
        hb, db = self.up.reserve(max(n, 16))
        h = hb[:n].view(dtype).reshape(shape)
        self.dev[name] = db[:n].view(_TORCH_OF[dtype]).view(tuple(shape))
        return h


class BatchInfo:
    """Stands where the reference passes the list of MolTree objects (``x_batch[0]``): the training-side
    callers only take its length, everything else they need is prepared with the holders."""

    def __init__(self, n: int, ids=None):
        self.n, self.ids = int(n), ids

    def __len__(self):
        return self.n


def _stub(rows: int, dtype=torch.int64) -> torch.Tensor:
    return torch.empty((int(rows), 0), dtype=dtype)


def _attach(t: torch.Tensor, tag, value) -> None:
    graphprep.cached_prep(t, tag, lambda: value)


def graph_prep(gs: flat.GraphSet, ids, device, mess=None, max_nb: int = flat.MPN_MAX_NB, train: bool = True,
               threads: int = None) -> GraphMPNPrep:
    """GraphMPNPrep of the selected graphs (``mess``: the batch's tree messages for candidate graphs)."""
    device = torch.device(device)
    with graphprep.packed_uploads(device):
        st = _Stage(device)
        o = flat.graph_csr(gs, ids, mess, max_nb, train, threads, alloc=st.alloc)
        d = st.dev
    na, rows, n_tree, n = o["n_atoms"], o["n_rows"], o["n_tree"], o["n_graphs"]

    def csr(rp, col, n_rows, nnz):
        return DeviceCSR(d[rp], d[col][:max(nnz, 1)], n_rows, nnz)

    n_src = n_tree + rows
    return GraphMPNPrep(  # the feature rows stay bytes; a dense product that wants floats widens them on demand
        fatoms=None, fbonds=None, fatoms_u8=d["fatoms"], fbonds_u8=d["fbonds"],
        atom_fdim=o["atom_fdim"], bond_in_fdim=o["bond_in_fdim"],
        n_atoms=na, n_bonds=rows, n_tree=n_tree,
        bcsr=csr("b_rowptr", "b_col", rows, o["b_nnz"]),
        bcsr_t=csr("bt_rowptr", "bt_col", n_src, o["b_nnz"]) if train else None,
        acsr=csr("a_rowptr", "a_col", na, o["a_nnz"]),
        acsr_t=csr("at_rowptr", "at_col", n_src, o["a_nnz"]) if train else None,
        seg=d["seg"], seg_of=d["seg_of"], n_seg=n, b_tree_nnz=o["b_tree_nnz"], a_tree_nnz=o["a_tree_nnz"],
        host=None)


def tree_prep(ts: flat.TreeSet, ids, device, vocab_size: int, depth: int, train: bool = True,
              all_messages: bool = False, threads: int = None, schedule: str = None):
    """(TreePrep, host arrays) of the selected trees with the forward (and backward) level schedules built;
    None instead of the TreePrep when a tree is deeper than ``depth`` iterations (the caller then takes the
    table route, which splits such trees off)."""
    from . import jtnn_enc

    device = torch.device(device)
    mode = schedule or jtnn_enc.SCHEDULE
    with graphprep.packed_uploads(device):
        st = _Stage(device)
        o = flat.tree_csr(ts, ids, threads, alloc=st.alloc)
        nn_, nm, n = o["n_nodes"], o["n_mess"], o["n_trees"]
        if nn_ == 0 or n == 0:
            raise ValueError("empty tree batch")
        if mode == "once" and o["max_sf"] > depth:
            return None, o
        fn, fm, roots = o["wid"][:nn_], o["src"], o["roots"][:n]
        graphprep.check_range(fn, vocab_size, "fnode")
        m_rp, m_col = o["m_rowptr"], o["m_col"][:o["m_nnz"]]
        n_rp, n_col = o["n_rowptr"], o["n_col"][:nm - 1]
        cnt = n_rp[roots + 1] - n_rp[roots]
        r_rp = np.zeros(n + 1, np.int32)
        np.cumsum(cnt, out=r_rp[1:])
        r_col = graphprep._expand_lists(n_rp, n_col, roots)
        mcsr = DeviceCSR(st.dev["m_rowptr"], st.dev["m_col"][:max(o["m_nnz"], 1)], nm, o["m_nnz"])
        prep = graphprep._tree_prep_from_csr(fn, fm, roots, m_rp, m_col, r_rp, r_col, device, vocab_size, o["sf"],
                                             dev_csr=(mcsr, graphprep.upload_csr(r_rp, r_col, device)))
    if train:
        prep.ensure_training()
    sched = graphprep.tree_schedule(prep, depth, mode=mode)
    if train:
        graphprep.backward_schedule(prep, sched, all_messages=all_messages)
    return prep, o


def _tree_holder(ts, ids, device, vocab_size, depth, train, all_messages, threads, want_plan=False):
    prep, o = tree_prep(ts, ids, device, vocab_size, depth, train, all_messages, threads)
    if prep is None:
        # a tree deeper than the iteration count: the reference's tables, through the modules' own preparation
        (fnode, fmess, node_graph, mess_graph, scope), (src, dst) = flat.tree_tables(ts, ids, threads)
        holder = (torch.from_numpy(fnode), torch.from_numpy(fmess), torch.from_numpy(node_graph),
                  torch.from_numpy(mess_graph), [(int(a), int(b)) for a, b in scope.tolist()])
        return holder, (src, dst, int(fnode.shape[0]))
    nn_, nm, n = o["n_nodes"], o["n_mess"], o["n_trees"]
    key = _stub(nm)
    _attach(key, ("deep_split", depth), None)
    _attach(key, ("tree", str(torch.device(device)), vocab_size), prep)
    scope = _LazyScope(ts, ids, n)
    if want_plan:
        from .jtnn_dec import decoder_plan_from_arrays

        _attach(key, ("decoder_plan",), decoder_plan_from_arrays(o["wid"][:nn_], o["src"], o["dst"], o["roots"][:n]))
    holder = (_stub(nn_), _stub(nm), _stub(nn_), key, scope)
    # src / dst sit in staging memory that is recycled once this batch is gone: the candidate builder reads them
    # right away, anyone who keeps them gets copies
    return holder, (o["src"], o["dst"], nn_)


class _LazyScope(Sequence):
    """``scope`` of a junction-tree holder ([(first node, nodes)] per tree), computed when somebody indexes it."""

    def __init__(self, ts, ids, n):
        self.ts, self.ids, self.n, self._v = ts, ids, int(n), None

    def _list(self):
        if self._v is None:
            g = np.arange(self.n) if self.ids is None else np.asarray(self.ids, dtype=np.int64)
            sizes = self.ts.node_ptr[g + 1] - self.ts.node_ptr[g]
            starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if self.n else sizes
            self._v = [(int(a), int(b)) for a, b in zip(starts, sizes)]
        return self._v

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self._list()[i]


def _graph_holder(prep: GraphMPNPrep, tag) -> tuple:
    key = _stub(prep.n_bonds)
    _attach(key, tag, prep)
    return (_stub(prep.n_atoms, torch.float32), _stub(prep.n_bonds, torch.float32), _stub(prep.n_atoms), key,
            _SegScope(prep))


class _SegScope(Sequence):
    """``scope`` of a graph holder, read back from the prepared segment table on demand."""

    def __init__(self, prep):
        self.prep, self._v = prep, None

    def __len__(self):
        return self.prep.n_seg

    def __getitem__(self, i):
        if self._v is None:
            self._v = [(int(a), int(b)) for a, b in self.prep.seg.cpu().tolist()]
        return self._v[i]


def _model_dims(model):
    return model.mpn.W_h.weight.device, int(model.jtnn.embedding.weight.shape[0]), int(model.jtnn.depth)


def encoder_batch(model, shard: flat.FlatShard, ids=None, train: bool = True, all_messages: bool = False,
                  threads: int = None):
    """(jtenc_holder, mpn_holder) of molecules ``ids`` of the shard, prepared on the model's device: what
    ``datautils.tensorize(..., assm=False)`` + the modules' own index preparation produce, without the tables."""
    device, V, depth = _model_dims(model)
    jt, _ = _tree_holder(shard.trees, ids, device, V, depth, train, all_messages, threads)
    mp = _graph_holder(graph_prep(shard.mols, ids, device, None, flat.MPN_MAX_NB, train, threads),
                       ("mpn", str(device), 0))
    return jt, mp


def training_batch(model, shard: flat.FlatShard, ids=None, props=None, threads: int = None):
    """The ``x_batch`` of ``JTNNVAE.forward`` (``(batch, jtenc_holder, mpn_holder, (jtmpn_holder, batch_idx))``,
    fast_jtnn/datautils.py:163) or, with ``props`` ([n, 2..4] property table), of ``JTpropVAE.forward``
    (fast_jtnn/datautils_prop.py:161-205), prepared on the model's device."""
    from .jtnn_dec import JTNNDecoder

    if shard.cands is None or shard.trees.cand_ptr is None:
        raise ValueError("the shard carries no attachment candidates")
    device, V, depth = _model_dims(model)
    n = shard.n if ids is None else int(len(ids))
    want_plan = isinstance(getattr(model, "decoder", None), JTNNDecoder)
    jt, (src, dst, n_nodes) = _tree_holder(shard.trees, ids, device, V, depth, True, True, threads, want_plan)
    mp = _graph_holder(graph_prep(shard.mols, ids, device, None, flat.MPN_MAX_NB, True, threads),
                       ("mpn", str(device), 0))
    cand_ids, batch_idx, node_base, seg_len, label_pos = flat.select_candidates(shard.trees, ids)
    n_tree = int(src.shape[0])
    mess = flat.candidate_mess(src, dst, n_nodes, node_base)
    cprep = graph_prep(shard.cands, cand_ids, device, mess, flat.JTMPN_MAX_NB, True, threads)
    cj = _graph_holder(cprep, ("jtmpn", str(device), n_tree))
    # what JTNNVAE.assm derives from batch_idx and from the tree objects, under the keys it looks them up by
    bi = _stub(batch_idx.shape[0])
    own_ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(batch_idx, minlength=n), out=own_ptr[1:])
    seg_ptr = np.zeros(seg_len.shape[0] + 1, np.int64)
    np.cumsum(seg_len, out=seg_ptr[1:])
    with graphprep.packed_uploads(device):
        owner = (graphprep._dev_i32(batch_idx, device), graphprep._dev_i32(own_ptr, device))
        seg = (graphprep._dev_i32(seg_ptr, device), graphprep._dev_i32(np.append(label_pos, 0), device))
    _attach(bi, ("assm_owner", str(device), n), owner)
    _attach(bi, ("assm_segments", str(device)), (seg[0], seg[1][:seg_len.shape[0]], int(seg_len.shape[0]),
                                                 int(seg_ptr[-1])))
    info = BatchInfo(n, ids)
    if props is not None:
        return info, props, jt, mp, (cj, bi)
    return info, jt, mp, (cj, bi)
