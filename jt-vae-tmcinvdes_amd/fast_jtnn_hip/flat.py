"""Flat storage of abstract graphs and the native tensorizers over it.

The reference keeps a data set as pickled ``MolTree`` objects and turns a batch of them into holder tables
with Python loops every time the batch is drawn (fast_jtnn/datautils.py:84-108, 140-163).  Here a data set
(or one batch) is a handful of flat arrays - atoms, bonds and one-hot feature bytes of the molecules, word ids
and neighbour lists of the junction trees, the attachment candidates of every tree node - and a batch is a
list of molecule positions in them.  ``libjtvae_hip.so`` builds from that, with a few host threads,

  * the reference's own tables, bit for bit (``*_tables``: what ``MPN.tensorize``, ``JTNNEncoder.tensorize``
    and ``JTMPN.tensorize`` return), or
  * directly what the kernels read (``*_csr``: byte feature rows, int32 CSR without padding slots, its
    transpose, segment tables, the settling iteration of every tree message), so that the padded int64 tables
    never exist.

Producing the flat arrays from SMILES is RDKit work (tree decomposition, candidate enumeration, feature
extraction: fast_jtnn/mol_tree.py, chemutils.py, mpn.py:46-68) and stays with the caller;
``graph_set_from_objects`` / ``tree_set_from_trees`` flatten objects that carry the same information.
"""
from __future__ import annotations

import ctypes as C
import os
from collections.abc import Mapping
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _capi as K
from .synthetic import JTMPN_MAX_NB, MPN_MAX_NB


def host_threads() -> int:
    """Host threads one tensorize / prepare call may use: the machine's cores divided among the ranks of the
    node, at most 8 (``JTVAE_HOST_THREADS`` overrides)."""
    env = os.environ.get("JTVAE_HOST_THREADS")
    if env:
        return max(1, int(env))
    world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    return max(1, min(8, (os.cpu_count() or 1) // world))


def _p(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data


def _arr(a, dtype, ndim=1, cols=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != ndim or (cols is not None and a.shape[-1] != cols):
        raise ValueError("expected a %d-d array%s, got shape %s" % (ndim, "" if cols is None else " with %d columns" % cols,
                                                                    a.shape))
    return a


def _raise(rc: int, what: str):
    if rc == K.E_RANGE:
        raise IndexError("%s: an index is out of range or a neighbour list exceeds the table width" % what)
    K.check(rc, what)


def _as_u8(x: np.ndarray, what: str) -> np.ndarray:
    """Feature rows as bytes; they are one-hot blocks (fast_jtnn/mpn.py:18-31), so nothing is lost - checked."""
    x = np.asarray(x)
    if x.dtype == np.uint8:
        return np.ascontiguousarray(x)
    u = x.astype(np.uint8)
    if not np.array_equal(u, x):
        raise ValueError("%s are not whole numbers in [0, 255]: the flat form stores one-hot rows as bytes" % what)
    return np.ascontiguousarray(u)


@dataclass
class GraphSet:
    """Small graphs back to back: molecules (MPN side) or attachment candidates (JTMPN side, with
    ``atom_nid``: the atom-map number of fast_jtnn/jtmpn.py:140)."""

    atom_ptr: np.ndarray  # int64 [n + 1]
    bond_ptr: np.ndarray  # int64 [n + 1]
    atom_feats: np.ndarray  # uint8 [atoms, atom_fdim]
    bonds: np.ndarray  # int32 [bonds, 2] (begin, end) positions inside the graph
    bond_feats: np.ndarray  # uint8 [bonds, bond_fdim]
    atom_nid: Optional[np.ndarray] = None  # int32 [atoms]

    def __post_init__(self):
        self.atom_ptr, self.bond_ptr = _arr(self.atom_ptr, np.int64), _arr(self.bond_ptr, np.int64)
        self.atom_feats, self.bond_feats = _arr(self.atom_feats, np.uint8, 2), _arr(self.bond_feats, np.uint8, 2)
        self.bonds = _arr(self.bonds, np.int32, 2, 2) if np.size(self.bonds) else np.zeros((0, 2), np.int32)
        if self.atom_nid is not None:
            self.atom_nid = _arr(self.atom_nid, np.int32)
        n = self.n
        if self.bond_ptr.shape[0] != n + 1 or n < 0:
            raise ValueError("atom_ptr / bond_ptr disagree on the number of graphs")
        if n >= 0 and (int(self.atom_ptr[-1]) != self.atom_feats.shape[0] or int(self.bond_ptr[-1]) != self.bonds.shape[0]
                       or self.bond_feats.shape[0] != self.bonds.shape[0]
                       or (self.atom_nid is not None and self.atom_nid.shape[0] != self.atom_feats.shape[0])):
            raise ValueError("array lengths do not match the pointer arrays")

    @property
    def n(self) -> int:
        return int(self.atom_ptr.shape[0]) - 1

    def c_struct(self) -> K.GraphSetC:
        return K.GraphSetC(self.n, _p(self.atom_ptr), _p(self.bond_ptr), _p(self.atom_feats), _p(self.bonds),
                           _p(self.bond_feats), _p(self.atom_nid), self.atom_feats.shape[1], self.bond_feats.shape[1])


@dataclass
class TreeSet:
    """Junction trees back to back: word ids and neighbour lists (positions inside the tree, in neighbour
    order; a tree's first node is its root) and, for training, the candidates of every node as a range of a
    candidate ``GraphSet`` plus the position of the true one."""

    node_ptr: np.ndarray  # int64 [n + 1]
    wid: np.ndarray  # int32 [nodes]
    nbr_ptr: np.ndarray  # int64 [nodes + 1]
    nbr: np.ndarray  # int32
    cand_ptr: Optional[np.ndarray] = None  # int64 [nodes + 1]
    label: Optional[np.ndarray] = None  # int32 [nodes], -1 = none

    def __post_init__(self):
        self.node_ptr, self.nbr_ptr = _arr(self.node_ptr, np.int64), _arr(self.nbr_ptr, np.int64)
        self.wid, self.nbr = _arr(self.wid, np.int32), _arr(self.nbr, np.int32)
        if self.cand_ptr is not None:
            self.cand_ptr, self.label = _arr(self.cand_ptr, np.int64), _arr(self.label, np.int32)
        nn = self.wid.shape[0]
        if int(self.node_ptr[-1]) != nn or self.nbr_ptr.shape[0] != nn + 1 or int(self.nbr_ptr[-1]) != self.nbr.shape[0] \
                or (self.cand_ptr is not None and (self.cand_ptr.shape[0] != nn + 1 or self.label.shape[0] != nn)):
            raise ValueError("array lengths do not match the pointer arrays")

    @property
    def n(self) -> int:
        return int(self.node_ptr.shape[0]) - 1

    def c_struct(self) -> K.TreeSetC:
        return K.TreeSetC(self.n, _p(self.node_ptr), _p(self.wid), _p(self.nbr_ptr), _p(self.nbr), _p(self.cand_ptr),
                          _p(self.label))


@dataclass
class FlatShard:
    """A data set (or one batch) in flat form: tree i belongs to molecule i."""

    trees: TreeSet
    mols: GraphSet
    cands: Optional[GraphSet] = None  # the graphs trees.cand_ptr points into

    @property
    def n(self) -> int:
        return self.trees.n


# ---- objects -> flat form -----------------------------------------------------------------------


def graph_set_from_objects(graphs: Sequence, with_nid: bool = False) -> GraphSet:
    """Objects with ``atom_feats [n, f]``, ``bonds [m, 2]``, ``bond_feats [m, g]`` (and ``atom_nid [n]``)."""
    na = np.fromiter((g.atom_feats.shape[0] for g in graphs), dtype=np.int64, count=len(graphs))
    nb = np.fromiter((g.bonds.shape[0] for g in graphs), dtype=np.int64, count=len(graphs))
    ap, bp = np.zeros(len(graphs) + 1, np.int64), np.zeros(len(graphs) + 1, np.int64)
    np.cumsum(na, out=ap[1:])
    np.cumsum(nb, out=bp[1:])
    if not graphs:
        raise ValueError("empty batch")
    fa = np.concatenate([g.atom_feats for g in graphs], axis=0)
    fb = np.concatenate([g.bond_feats for g in graphs], axis=0)
    bonds = np.concatenate([np.asarray(g.bonds).reshape(-1, 2) for g in graphs], axis=0)
    nid = np.concatenate([g.atom_nid for g in graphs]) if with_nid else None
    return GraphSet(ap, bp, _as_u8(fa, "atom features"), bonds, _as_u8(fb, "bond features"), nid)


def tree_set_from_nodes(node_batch: Sequence, tree_sizes: Sequence[int], with_cands: bool = False):
    """Nodes in batch order with ``wid``, ``neighbors`` and ``idx`` (fast_jtnn/datautils.py:166-172: the node's
    position in the batch); ``tree_sizes``: nodes per tree.  With ``with_cands`` also the candidates
    (``node.cands``, ``node.label``) as a GraphSet: returns (TreeSet, GraphSet | None)."""
    nn = len(node_batch)
    sizes = np.asarray(tree_sizes, dtype=np.int64)
    node_ptr = np.zeros(sizes.shape[0] + 1, np.int64)
    np.cumsum(sizes, out=node_ptr[1:])
    if int(node_ptr[-1]) != nn:
        raise ValueError("tree sizes do not add up to the number of nodes")
    wid = np.fromiter((x.wid for x in node_batch), dtype=np.int32, count=nn)
    deg = np.fromiter((len(x.neighbors) for x in node_batch), dtype=np.int64, count=nn)
    nbr_ptr = np.zeros(nn + 1, np.int64)
    np.cumsum(deg, out=nbr_ptr[1:])
    idx = np.fromiter((x.idx for x in node_batch), dtype=np.int64, count=nn)
    if not np.array_equal(idx, np.arange(nn)):
        raise ValueError("node.idx must be the node's position in the batch (set_batch_nodeID)")
    nbr = np.fromiter((y.idx for x in node_batch for y in x.neighbors), dtype=np.int64, count=int(nbr_ptr[-1]))
    first = np.repeat(np.repeat(node_ptr[:-1], sizes), deg)  # first node of the tree each list entry sits in
    ts = TreeSet(node_ptr, wid, nbr_ptr, (nbr - first).astype(np.int32))
    if not with_cands:
        return ts, None
    cands, cand_ptr, label = [], np.zeros(nn + 1, np.int64), np.full(nn, -1, np.int32)
    for j, x in enumerate(node_batch):
        cs = x.cands or ()
        cands.extend(cs)
        cand_ptr[j + 1] = len(cands)
        if x.label is not None and len(cs):
            label[j] = cs.index(x.label)
    ts.cand_ptr, ts.label = cand_ptr, label
    gs = graph_set_from_objects(cands, with_nid=True) if cands else None
    return ts, gs


def shard_from_trees(tree_batch: Sequence, with_cands: bool = False) -> FlatShard:
    """Tree objects (``nodes``, ``mol``; node ids set by ``set_batch_nodeID``) -> flat form."""
    nodes = [x for t in tree_batch for x in t.nodes]
    ts, cs = tree_set_from_nodes(nodes, [len(t.nodes) for t in tree_batch], with_cands)
    return FlatShard(ts, graph_set_from_objects([t.mol for t in tree_batch]), cs)


# ---- flat form on disk ------------------------------------------------------------------------------


def save_shard(path: str, shard: FlatShard, props: Optional[np.ndarray] = None) -> None:
    """One ``.npz`` of plain arrays (no pickled objects): what the reference keeps as pickled MolTree lists
    (fast_molvae/preprocess.py:39-57).  ``props``: the property table of the conditional model, one row per
    molecule (fast_jtnn/datautils_prop.py:161-175)."""
    a = {}
    for prefix, obj, names in (("t_", shard.trees, ("node_ptr", "wid", "nbr_ptr", "nbr", "cand_ptr", "label")),
                               ("m_", shard.mols, ("atom_ptr", "bond_ptr", "atom_feats", "bonds", "bond_feats")),
                               ("c_", shard.cands, ("atom_ptr", "bond_ptr", "atom_feats", "bonds", "bond_feats",
                                                    "atom_nid"))):
        if obj is None:
            continue
        for n in names:
            v = getattr(obj, n)
            if v is not None:
                a[prefix + n] = v
    if props is not None:
        a["props"] = np.asarray(props, dtype=np.float32)
    np.savez(path, **a)


def load_shard(path: str):
    """(FlatShard, props or None) of a file written by ``save_shard``; nothing in the file is executed."""
    z = np.load(path, allow_pickle=False)
    g = lambda k: z[k] if k in z.files else None  # noqa: E731
    trees = TreeSet(z["t_node_ptr"], z["t_wid"], z["t_nbr_ptr"], z["t_nbr"], g("t_cand_ptr"), g("t_label"))
    mols = GraphSet(z["m_atom_ptr"], z["m_bond_ptr"], z["m_atom_feats"], z["m_bonds"], z["m_bond_feats"])
    cands = None
    if "c_atom_ptr" in z.files:
        cands = GraphSet(z["c_atom_ptr"], z["c_bond_ptr"], z["c_atom_feats"], z["c_bonds"], z["c_bond_feats"],
                         z["c_atom_nid"])
    return FlatShard(trees, mols, cands), g("props")


# ---- the mess_dict of JTNNEncoder.tensorize, backed by arrays --------------------------------------


class MessDict(Mapping):
    """``{(x.idx, y.idx): message id}`` (fast_jtnn/jtnn_enc.py:58-62) over the src / dst arrays of the batch's
    messages; the Python dictionary is only built if somebody iterates or looks a pair up."""

    def __init__(self, src: np.ndarray, dst: np.ndarray, n_nodes: int):
        self.src, self.dst, self.n_nodes = src, dst, int(n_nodes)  # int32 [n_mess], entry 0 = padding message
        self._d = None

    def _dict(self):
        if self._d is None:
            self._d = {(int(a), int(b)): i + 1 for i, (a, b) in enumerate(zip(self.src[1:].tolist(), self.dst[1:].tolist()))}
        return self._d

    def __len__(self):
        return int(self.src.shape[0]) - 1 if self._d is None else len(self._d)

    def __getitem__(self, k):
        return self._dict()[k]

    def __iter__(self):
        return iter(self._dict())

    def __contains__(self, k):
        return k in self._dict()


def mess_arrays(mess_dict, n_nodes: Optional[int] = None):
    """(src, dst int32 [n_mess], n_nodes) of a mess_dict: the arrays of a MessDict, or a plain dictionary laid
    out by id (ids must be 1 .. len)."""
    if isinstance(mess_dict, MessDict):
        return mess_dict.src, mess_dict.dst, mess_dict.n_nodes
    n = len(mess_dict)
    src, dst = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    if n:
        keys = np.asarray(list(mess_dict.keys()), dtype=np.int64).reshape(-1, 2)
        ids = np.fromiter(mess_dict.values(), dtype=np.int64, count=n)
        if ids.min() < 1 or ids.max() > n or np.unique(ids).shape[0] != n:
            raise ValueError("mess_dict ids must be 1 .. %d, each once" % n)
        src[ids], dst[ids] = keys[:, 0], keys[:, 1]
    if n_nodes is None:
        n_nodes = int(max(src.max(), dst.max())) + 1 if n else 0
    return src, dst, int(n_nodes)


# ---- native tensorizers -------------------------------------------------------------------------


def _ids(ids, n):
    if ids is None:
        return None, n
    ids = _arr(ids, np.int64)
    return ids, int(ids.shape[0])


class _Mess:
    def __init__(self, src, dst, n_nodes, node_base):
        self.src, self.dst = _arr(src, np.int32), _arr(dst, np.int32)
        self.node_base = _arr(node_base, np.int64)
        self.c = K.MessIndexC(self.src.shape[0], int(n_nodes), self.src.ctypes.data, self.dst.ctypes.data,
                              _p(self.node_base))


def graph_counts(gs: GraphSet, ids=None, mess: Optional[_Mess] = None):
    ids, n = _ids(ids, gs.n)
    counts = np.zeros(4, np.int64)
    c = gs.c_struct()
    _raise(K.lib.jtvae_tensorize_graph_counts(C.addressof(c), _p(ids), n, None if mess is None else C.addressof(mess.c),
                                              counts.ctypes.data), "tensorize_graph_counts")
    return tuple(int(v) for v in counts)  # atoms, bonds, a_cap, b_cap


def graph_tables(gs: GraphSet, ids=None, mess: Optional[_Mess] = None, max_nb: int = MPN_MAX_NB, threads: int = None):
    """(fatoms f32, fbonds f32, agraph i64, bgraph i64, scope i64 [n, 2]) in the reference's layout."""
    ids, n = _ids(ids, gs.n)
    if mess is not None and mess.node_base.shape[0] != n:
        raise ValueError("node_base must have one entry per selected graph")
    na, nb, _, _ = graph_counts(gs, ids, mess)
    rows = 2 * nb + (0 if mess is not None else 1)
    fa, fb = gs.atom_feats.shape[1], gs.bond_feats.shape[1]
    out = (np.empty((na, fa), np.float32), np.empty((rows, fa + fb), np.float32), np.empty((na, max_nb), np.int64),
           np.empty((rows, max_nb), np.int64), np.empty((n, 2), np.int64))
    tb = K.GraphTablesC(*[_p(a) for a in out])
    c = gs.c_struct()
    _raise(K.lib.jtvae_tensorize_graphs(C.addressof(c), _p(ids), n, max_nb, None if mess is None else C.addressof(mess.c),
                                        C.addressof(tb), None, threads or host_threads()), "tensorize_graphs")
    return out


def graph_csr(gs: GraphSet, ids=None, mess: Optional[_Mess] = None, max_nb: int = MPN_MAX_NB, train: bool = True,
              threads: int = None, alloc=None):
    """The device-ready form as a dict of host arrays (``alloc(name, shape, dtype)`` provides the memory, e.g.
    slices of a pinned staging buffer; default: numpy)."""
    ids, n = _ids(ids, gs.n)
    if mess is not None and mess.node_base.shape[0] != n:
        raise ValueError("node_base must have one entry per selected graph")
    na, nb, a_cap, b_cap = graph_counts(gs, ids, mess)
    n_tree = 0 if mess is None else int(mess.src.shape[0])
    rows = 2 * nb + (0 if mess is not None else 1)
    n_src = n_tree + rows
    fa, fb = gs.atom_feats.shape[1], gs.bond_feats.shape[1]
    alloc = alloc or (lambda name, shape, dtype: np.empty(shape, dtype))
    o = dict(fatoms=alloc("fatoms", (na, fa), np.uint8), fbonds=alloc("fbonds", (rows, fa + fb), np.uint8),
             a_rowptr=alloc("a_rowptr", (na + 1,), np.int32), a_col=alloc("a_col", (max(a_cap, 1),), np.int32),
             b_rowptr=alloc("b_rowptr", (rows + 1,), np.int32), b_col=alloc("b_col", (max(b_cap, 1),), np.int32),
             seg=alloc("seg", (n, 2), np.int32), seg_of=alloc("seg_of", (max(na, 1),), np.int32))
    if train:
        o.update(at_rowptr=alloc("at_rowptr", (n_src + 1,), np.int32), at_col=alloc("at_col", (max(a_cap, 1),), np.int32),
                 bt_rowptr=alloc("bt_rowptr", (n_src + 1,), np.int32), bt_col=alloc("bt_col", (max(b_cap, 1),), np.int32))
    g = lambda k: o[k].ctypes.data if k in o else None  # noqa: E731
    cs = K.GraphCsrC(g("fatoms"), g("fbonds"), g("a_rowptr"), g("a_col"), g("b_rowptr"), g("b_col"), a_cap, b_cap,
                     g("at_rowptr"), g("at_col"), g("bt_rowptr"), g("bt_col"), g("seg"), g("seg_of"))
    c = gs.c_struct()
    _raise(K.lib.jtvae_tensorize_graphs(C.addressof(c), _p(ids), n, max_nb, None if mess is None else C.addressof(mess.c),
                                        None, C.addressof(cs), threads or host_threads()), "tensorize_graphs")
    a_nnz, b_nnz, a_tree, b_tree = (int(v) for v in cs.nnz)
    o.update(n_atoms=na, n_rows=rows, n_tree=n_tree, n_graphs=n, a_nnz=a_nnz, b_nnz=b_nnz, a_tree_nnz=a_tree,
             b_tree_nnz=b_tree, atom_fdim=fa, bond_in_fdim=fa + fb)
    return o


def tree_counts(ts: TreeSet, ids=None, widths: bool = True, threads: int = None):
    ids, n = _ids(ids, ts.n)
    counts = np.zeros(5, np.int64)
    c = ts.c_struct()
    _raise(K.lib.jtvae_tensorize_tree_counts(C.addressof(c), _p(ids), n, int(widths), threads or host_threads(),
                                             counts.ctypes.data), "tensorize_tree_counts")
    return tuple(int(v) for v in counts)  # nodes, messages, Kn, Km, m_cap


def tree_tables(ts: TreeSet, ids=None, threads: int = None):
    """((fnode, fmess, node_graph, mess_graph int64, scope int64 [n, 2]), (src, dst int32)) in the reference's
    layout; src / dst are the arrays behind the mess_dict."""
    ids, n = _ids(ids, ts.n)
    nn, nm, kn, km, m_cap = tree_counts(ts, ids)
    out = (np.empty(nn, np.int64), np.empty(nm, np.int64), np.empty((nn, kn), np.int64), np.empty((nm, km), np.int64),
           np.empty((n, 2), np.int64))
    tb = K.TreeTablesC(*[_p(a) for a in out], kn, km)
    aux = dict(wid=np.empty(max(nn, 1), np.int32), src=np.empty(nm, np.int32), dst=np.empty(nm, np.int32),
               n_rowptr=np.empty(nn + 1, np.int32), n_col=np.empty(max(nm - 1, 1), np.int32),
               m_rowptr=np.empty(nm + 1, np.int32), m_col=np.empty(max(m_cap, 1), np.int32))
    cs = K.TreeCsrC(*[aux[k].ctypes.data for k in ("wid", "src", "dst", "n_rowptr", "n_col", "m_rowptr", "m_col")], m_cap,
                    None, None)
    c = ts.c_struct()
    _raise(K.lib.jtvae_tensorize_trees(C.addressof(c), _p(ids), n, C.addressof(tb), C.addressof(cs),
                                       threads or host_threads()), "tensorize_trees")
    return out, (aux["src"], aux["dst"])


def tree_csr(ts: TreeSet, ids=None, threads: int = None, alloc=None):
    """The device-ready form of the junction trees as a dict of host arrays (see ``graph_csr`` for ``alloc``)."""
    ids, n = _ids(ids, ts.n)
    nn, nm, _, _, m_cap = tree_counts(ts, ids, widths=False, threads=threads)
    alloc = alloc or (lambda name, shape, dtype: np.empty(shape, dtype))
    o = dict(wid=alloc("wid", (max(nn, 1),), np.int32), src=alloc("src", (nm,), np.int32), dst=alloc("dst", (nm,), np.int32),
             n_rowptr=alloc("n_rowptr", (nn + 1,), np.int32), n_col=alloc("n_col", (max(nm - 1, 1),), np.int32),
             m_rowptr=alloc("m_rowptr", (nm + 1,), np.int32), m_col=alloc("m_col", (max(m_cap, 1),), np.int32),
             sf=alloc("sf", (nm,), np.int32), roots=alloc("roots", (max(n, 1),), np.int32))
    cs = K.TreeCsrC(*[o[k].ctypes.data for k in ("wid", "src", "dst", "n_rowptr", "n_col", "m_rowptr", "m_col")], m_cap,
                    o["sf"].ctypes.data, o["roots"].ctypes.data)
    c = ts.c_struct()
    _raise(K.lib.jtvae_tensorize_trees(C.addressof(c), _p(ids), n, None, C.addressof(cs), threads or host_threads()),
           "tensorize_trees")
    o.update(n_nodes=nn, n_mess=nm, n_trees=n, m_nnz=int(cs.nnz), max_sf=int(cs.max_sf))
    return o


def select_candidates(ts: TreeSet, ids=None):
    """(cand_ids int64, batch_idx int32, node_base int64, seg_len int32, label_pos int32): the candidate list of
    fast_jtnn/datautils.py:150-161 and the loss segments of fast_jtnn/jtnn_vae.py:116-135."""
    if ts.cand_ptr is None:
        raise ValueError("the tree set carries no candidates")
    ids, n = _ids(ids, ts.n)
    counts = np.zeros(2, np.int64)
    c = ts.c_struct()
    _raise(K.lib.jtvae_select_candidates_counts(C.addressof(c), _p(ids), n, counts.ctypes.data), "select_candidates")
    nc, ns = int(counts[0]), int(counts[1])
    out = (np.empty(nc, np.int64), np.empty(nc, np.int32), np.empty(nc, np.int64), np.empty(ns, np.int32),
           np.empty(ns, np.int32))
    _raise(K.lib.jtvae_select_candidates(C.addressof(c), _p(ids), n, *[_p(a) for a in out]), "select_candidates")
    return out


def candidate_mess(src, dst, n_nodes, node_base) -> _Mess:
    return _Mess(src, dst, n_nodes, node_base)


__all__ = ["GraphSet", "TreeSet", "FlatShard", "MessDict", "graph_set_from_objects", "tree_set_from_nodes",
           "shard_from_trees", "save_shard", "load_shard", "graph_tables", "graph_csr", "tree_tables", "tree_csr", "select_candidates",
           "candidate_mess", "mess_arrays", "host_threads", "JTMPN_MAX_NB", "MPN_MAX_NB"]
