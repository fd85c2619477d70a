"""Per-batch index preparation: the reference's padded neighbour tables -> device CSR.

The reference hands the encoder padded int64 tables (agraph/bgraph [N,6] or [N,15],
node_graph/mess_graph [N,Kmax]) in which 0 means "no neighbour"
(fast_jtnn/mpn.py:148-159, fast_jtnn/jtnn_enc.py:78-90, fast_jtnn/jtmpn.py:170-181).
The kernels want, per batch and once for all 2 / 20 iterations:

  * the same adjacency as int32 CSR without the padding slots (slot order kept,
    so sums run in the reference's neighbour order),
  * its transpose (who reads row j), which turns the backward scatter-add into a
    second gather: no atomics, bitwise reproducible,
  * small derived index vectors (word id per message, molecule of each atom ...).

This is index bookkeeping in the same class as tensorize itself; it runs on the
host once per batch and its result is cached on the holder tuple.  The passes over
whole tables (CSR conversion, transposition, stable-iteration levels, grouping by
key) are native single-pass routines of the C library; the ``*_np`` functions are
their numpy statements, kept as the checkers of the native ones.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi as K


@dataclass
class DeviceCSR:
    rowptr: torch.Tensor  # int32 [n_rows + 1]
    col: torch.Tensor  # int32 [nnz]
    n_rows: int
    nnz: int


def _to_np(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def padded_to_csr(table: np.ndarray, n_src: int = 2 ** 31 - 1) -> Tuple[np.ndarray, np.ndarray]:
    """Rows of a zero-padded table -> (rowptr, col) keeping the slot order of the non-zeros.
    Entries outside [0, n_src) raise IndexError."""
    return K.prep_padded_to_csr(table, n_src)


def transpose_csr(rowptr: np.ndarray, col: np.ndarray, n_src: int) -> Tuple[np.ndarray, np.ndarray]:
    """For every source row j the list of rows i that read it, ordered by (i, slot)."""
    return K.prep_transpose_csr(rowptr, col, n_src)


def padded_to_csr_np(table: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """numpy statement of padded_to_csr."""
    table = np.ascontiguousarray(table)
    if table.ndim != 2:
        raise ValueError("neighbour table must be 2-D, got shape %s" % (table.shape,))
    nz = table != 0
    counts = nz.sum(axis=1).astype(np.int64)
    rowptr = np.zeros(table.shape[0] + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = table[nz]  # row-major order: row by row, slot by slot
    return rowptr, col.astype(np.int64)


def transpose_csr_np(rowptr: np.ndarray, col: np.ndarray, n_src: int) -> Tuple[np.ndarray, np.ndarray]:
    """numpy statement of transpose_csr."""
    n_rows = rowptr.shape[0] - 1
    counts = np.diff(rowptr)
    row_of = np.repeat(np.arange(n_rows, dtype=np.int64), counts)
    order = np.argsort(col, kind="stable")
    tcol = row_of[order]
    trowptr = np.zeros(n_src + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=n_src), out=trowptr[1:])
    return trowptr, tcol


def chunk_lists(rowptr: np.ndarray, chunk: int) -> Tuple[np.ndarray, np.ndarray]:
    """Split every list into pieces of at most ``chunk`` entries.

    Returns (piece_rowptr over the same col array, key_rowptr: pieces of each key).
    Long lists (all messages that share a word id) then sum in parallel, piece by
    piece, and the pieces of a key are added in order.
    """
    rowptr = np.asarray(rowptr, dtype=np.int64)
    counts = np.diff(rowptr)
    n_pieces = np.maximum(1, -(-counts // chunk))  # an empty list keeps one empty piece
    key_rowptr = np.zeros(rowptr.shape[0], dtype=np.int64)
    np.cumsum(n_pieces, out=key_rowptr[1:])
    total = int(key_rowptr[-1])
    key_of_piece = np.repeat(np.arange(counts.shape[0], dtype=np.int64), n_pieces)
    idx_in_key = np.arange(total, dtype=np.int64) - key_rowptr[key_of_piece]
    start = rowptr[key_of_piece] + idx_in_key * chunk
    end = np.minimum(start + chunk, rowptr[key_of_piece + 1])
    piece_rowptr = np.empty(total + 1, dtype=np.int64)
    piece_rowptr[:-1] = start
    piece_rowptr[-1] = rowptr[-1]
    # pieces are laid out in key order and lists are contiguous, so start[i+1] == end[i]
    assert total == 0 or np.array_equal(piece_rowptr[1:], end)
    return piece_rowptr, key_rowptr


def check_range(col: np.ndarray, n_src: int, what: str) -> None:
    if col.size and (int(col.min()) < 0 or int(col.max()) >= n_src):
        raise IndexError("%s: neighbour index out of range [0, %d): min %d max %d"
                         % (what, n_src, int(col.min()), int(col.max())))


class _UploadBatch:
    """Index arrays of one prepared holder, shipped with one host-to-device copy per 4 MB chunk instead of
    one small copy each (a tree holder with its schedules is ~100 arrays).  Arrays are laid out in pinned
    staging memory as they are added and get their device views at once; ``flush`` issues the copies on the
    current stream, ahead of every kernel that will read them."""

    CHUNK = 1 << 20  # int32 elements

    def __init__(self, device):
        self.device = torch.device(device)
        self.chunks = []  # [host, dev, used]

    def add(self, a: np.ndarray) -> torch.Tensor:
        n = int(a.size)
        n4 = (n + 3) & ~3  # keep every array 16-byte aligned
        if not self.chunks or self.chunks[-1][2] + n4 > self.chunks[-1][0].numel():
            size = max(self.CHUNK, n4)
            self.chunks.append([torch.empty(size, dtype=torch.int32, pin_memory=True),
                                _new_device_tensor(torch.empty(size, dtype=torch.int32, device=self.device)), 0])
        host, dev, used = self.chunks[-1]
        host.numpy()[used:used + n] = a.reshape(-1)
        self.chunks[-1][2] = used + n4
        return dev[used:used + n]

    def reserve(self, n_bytes: int):
        """(host bytes, device bytes) of a 16-byte aligned block of the staging memory that the caller fills in
        place before ``flush`` (the native builders write their results straight into it)."""
        n4 = (int(n_bytes) + 15) // 16 * 4
        if not self.chunks or self.chunks[-1][2] + n4 > self.chunks[-1][0].numel():
            size = max(self.CHUNK, n4)
            self.chunks.append([torch.empty(size, dtype=torch.int32, pin_memory=True),
                                _new_device_tensor(torch.empty(size, dtype=torch.int32, device=self.device)), 0])
        host, dev, used = self.chunks[-1]
        self.chunks[-1][2] = used + n4
        return host.numpy()[used:used + n4].view(np.uint8)[:n_bytes], dev[used:used + n4].view(torch.uint8)[:n_bytes]

    def flush(self):
        for host, dev, used in self.chunks:
            if used:
                dev[:used].copy_(host[:used], non_blocking=True)
        self.chunks = []


class _ThreadState(threading.local):
    upload = None  # the _UploadBatch of the innermost packed_uploads block of this thread
    consumer_stream = None  # set by a prefetch thread: the stream whose kernels will read what it uploads


_TLS = _ThreadState()


def _new_device_tensor(t: torch.Tensor) -> torch.Tensor:
    """Every device tensor this module allocates passes through here.  A prefetch thread allocates and fills
    on its own copy stream; telling the allocator which stream will read the tensor keeps the memory from
    being handed out again while those reads are still queued."""
    if _TLS.consumer_stream is not None and t.is_cuda:
        t.record_stream(_TLS.consumer_stream)
    return t


class consumer_stream:
    """``with consumer_stream(s):`` - device tensors created inside (on this thread) will be read on ``s``."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.prev = _TLS.consumer_stream
        _TLS.consumer_stream = self.stream
        return self

    def __exit__(self, *exc):
        _TLS.consumer_stream = self.prev
        return False


class packed_uploads:
    """``with packed_uploads(device):`` - index arrays created inside go to the device in one copy."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        self.prev = _TLS.upload
        _TLS.upload = _UploadBatch(self.device) if self.device.type == "cuda" else None
        return self

    def __exit__(self, *exc):
        if _TLS.upload is not None:
            _TLS.upload.flush()
        _TLS.upload = self.prev
        return False


def _dev_i32(a: np.ndarray, device) -> torch.Tensor:
    if a.dtype != np.int32:  # an int32 array cannot overflow the conversion: no scan
        if a.size and int(a.max()) >= 2 ** 31:
            raise OverflowError("index exceeds int32")
        a = a.astype(np.int32)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    if _TLS.upload is not None and _TLS.upload.device == torch.device(device):
        return _TLS.upload.add(a)
    return _new_device_tensor(torch.from_numpy(a).to(device))


def upload_csr(rowptr: np.ndarray, col: np.ndarray, device) -> DeviceCSR:
    col_t = _dev_i32(col if col.size else np.zeros(1, dtype=np.int64), device)
    return DeviceCSR(_dev_i32(rowptr, device), col_t, int(rowptr.shape[0] - 1), int(col.shape[0]))


def _widen_features(packed: torch.Tensor) -> torch.Tensor:
    """[n, f] bytes on the device -> [n, round4(f)] float rows with zero padding columns (one kernel)."""
    n, f = packed.shape
    fp = (f + 3) // 4 * 4
    out = _new_device_tensor(torch.empty((n, fp), dtype=torch.float32, device=packed.device))
    if n:
        if packed.is_cuda:
            with torch.cuda.device(packed.device):
                K.widen_u8_rows(packed.data_ptr(), out.data_ptr(), fp, n, f)
        else:  # CPU stand-in for tests of the index side
            out.zero_()
            out[:, :f] = packed.float()
    return out


def _upload_features(x: np.ndarray, device):
    """[n, f] float32 feature rows -> (float rows [n, round4(f)] or None, byte rows [n, f] or None) on the device.

    The rows are one-hot blocks (fast_jtnn/mpn.py:18-31).  When every value is a whole number in [0, 255] -
    checked in the pass that converts them - they cross the link as bytes in pinned memory, a quarter of the
    float size, and stay bytes: the feature layers read them as they are (``jtvae_onehot_linear``), and a consumer
    that wants float rows widens them on the device (``GraphMPNPrep.float_rows``: the same bits as shipping the
    floats).  Any other content goes as floats and is padded on the device."""
    n, f = x.shape
    fp = (f + 3) // 4 * 4
    x = np.ascontiguousarray(x)
    device = torch.device(device)
    if n and device.type == "cuda":
        stage = torch.empty((n, f), dtype=torch.uint8, pin_memory=True)
        if K.prep_features_to_u8(x, stage.numpy()):
            return None, _new_device_tensor(stage.to(device, non_blocking=True))
    src = _new_device_tensor(torch.from_numpy(x).to(device, non_blocking=True))
    if fp == f:
        return src, None
    out = _new_device_tensor(torch.zeros((n, fp), dtype=torch.float32, device=device))
    out[:, :f] = src
    return out, None


def scope_arrays(scope: Sequence[Tuple[int, int]], n_rows: int) -> Tuple[np.ndarray, np.ndarray]:
    seg = np.asarray(scope, dtype=np.int64).reshape(-1, 2)
    if seg.size and (seg[:, 0].min() < 0 or (seg[:, 0] + seg[:, 1]).max() > n_rows or seg[:, 1].min() < 0):
        raise IndexError("scope exceeds the %d rows it indexes" % n_rows)
    lens = seg[:, 1] if seg.size else np.zeros(0, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]) if seg.size else lens
    if seg.size and np.array_equal(seg[:, 0], starts) and int(lens.sum()) == n_rows:
        seg_of = np.repeat(np.arange(seg.shape[0], dtype=np.int64), lens)  # the tensorizers' layout
    else:
        seg_of = np.full(n_rows, -1, dtype=np.int64)  # rows outside every segment get no gradient
        for b in range(seg.shape[0]):  # any other disjoint layout
            seg_of[seg[b, 0]:seg[b, 0] + seg[b, 1]] = b
    return seg, seg_of


@dataclass
class GraphMPNPrep:
    """Device-side form of an (fatoms, fbonds, agraph, bgraph, scope) holder.

    Source rows of both gathers live in one index space: ``n_tree`` tree-message
    rows first (0 for the plain MPN), then the ``n_bonds`` bond-message rows.
    """

    fatoms: Optional[torch.Tensor]  # [NA, round4(atom_fdim)] float rows, or None while only the bytes exist
    fbonds: Optional[torch.Tensor]  # [NB, round4(atom_fdim + bond_fdim)]
    atom_fdim: int
    bond_in_fdim: int
    n_atoms: int
    n_bonds: int
    n_tree: int
    bcsr: DeviceCSR
    bcsr_t: DeviceCSR  # n_tree + n_bonds rows
    acsr: DeviceCSR
    acsr_t: DeviceCSR
    seg: torch.Tensor  # int32 [B, 2]
    seg_of: torch.Tensor  # int32 [NA]
    n_seg: int
    b_tree_nnz: int = 0  # bgraph / agraph entries that point at tree-message rows
    a_tree_nnz: int = 0
    host: dict = None  # CSR arrays kept for the transposes when they were not built (inference)
    fatoms_u8: Optional[torch.Tensor] = None  # [NA, atom_fdim] bytes when every feature is a whole number 0..255
    fbonds_u8: Optional[torch.Tensor] = None  # [NB, atom_fdim + bond_fdim]

    def float_rows(self, which: str) -> torch.Tensor:
        """``fatoms`` / ``fbonds`` as padded float rows (for the dense products), widened from the bytes on first use."""
        rows = getattr(self, which)
        if rows is None:
            rows = _widen_features(getattr(self, which + "_u8"))
            setattr(self, which, rows)
        return rows

    def ensure_transposes(self):
        """The transposed adjacency is only read by the backward pass; a holder prepared for inference gets it
        on its first use under gradient mode."""
        if self.bcsr_t is None:
            h, dev = self.host, self.seg.device
            n_src = self.n_tree + self.n_bonds
            with packed_uploads(dev):
                self.bcsr_t = upload_csr(*transpose_csr(h["b_rp"], h["b_col"], n_src), dev)
                self.acsr_t = upload_csr(*transpose_csr(h["a_rp"], h["a_col"], n_src), dev)
        return self


def prepare_mpn(holder, device, n_tree: int = 0, train: bool = True) -> GraphMPNPrep:
    with packed_uploads(device):
        return _prepare_mpn(holder, device, n_tree, train)


def _prepare_mpn(holder, device, n_tree: int = 0, train: bool = True) -> GraphMPNPrep:
    fatoms, fbonds, agraph, bgraph, scope = holder
    fa, fb = np.asarray(_to_np(fatoms), dtype=np.float32), np.asarray(_to_np(fbonds), dtype=np.float32)
    ag, bg = _to_np(agraph), _to_np(bgraph)
    na, nb = fa.shape[0], fb.shape[0]
    if ag.shape[0] != na or bg.shape[0] != nb:
        raise ValueError("agraph/bgraph row counts (%d, %d) do not match fatoms/fbonds (%d, %d)"
                         % (ag.shape[0], bg.shape[0], na, nb))
    n_src = n_tree + nb
    b_rp, b_col = padded_to_csr(bg, n_src)
    a_rp, a_col = padded_to_csr(ag, n_src)
    seg, seg_of = scope_arrays(scope, na)
    fa_f, fa_u = _upload_features(fa, device)
    fb_f, fb_u = _upload_features(fb, device)
    return GraphMPNPrep(
        fatoms=fa_f, fbonds=fb_f, fatoms_u8=fa_u, fbonds_u8=fb_u,
        atom_fdim=fa.shape[1], bond_in_fdim=fb.shape[1],
        n_atoms=na, n_bonds=nb, n_tree=n_tree,
        bcsr=upload_csr(b_rp, b_col, device),
        bcsr_t=upload_csr(*transpose_csr(b_rp, b_col, n_src), device) if train else None,
        acsr=upload_csr(a_rp, a_col, device),
        acsr_t=upload_csr(*transpose_csr(a_rp, a_col, n_src), device) if train else None,
        host=None if train else dict(b_rp=b_rp, b_col=b_col, a_rp=a_rp, a_col=a_col),
        seg=_dev_i32(seg.reshape(-1), device).view(-1, 2) if seg.size else torch.zeros((0, 2), dtype=torch.int32, device=device),
        seg_of=_dev_i32(seg_of, device) if na else torch.zeros(1, dtype=torch.int32, device=device),
        n_seg=int(seg.shape[0]),
        b_tree_nnz=int((b_col < n_tree).sum()), a_tree_nnz=int((a_col < n_tree).sum()),
    )


@dataclass
class TreePrep:
    """Device-side form of an (fnode, fmess, node_graph, mess_graph, scope) holder."""

    n_nodes: int
    n_mess: int
    n_trees: int
    n_words: int  # distinct word ids in the batch
    words: torch.Tensor  # int32 [n_words] word id of each table row
    word_csr: DeviceCSR  # identity-shaped CSR: row u gathers embedding row words[u]
    word_csr_t: DeviceCSR  # over the full vocabulary: which table row (if any) a word maps to
    wid_m: torch.Tensor  # int32 [NM] table row of the source node of each message
    wid_root: torch.Tensor  # int32 [B]
    mcsr: DeviceCSR
    mcsr_t: DeviceCSR
    rcsr: DeviceCSR  # B rows: incoming messages of each root node
    rcsr_t: DeviceCSR  # NM rows: the roots a message flows into
    ident: DeviceCSR  # B-row identity
    key_m_pieces: DeviceCSR  # messages grouped by table row, cut in pieces
    key_m: DeviceCSR  # pieces of each table row
    key_r_pieces: DeviceCSR
    key_r: DeviceCSR
    vocab_size: int
    host: dict = None  # numpy copies needed to build schedules lazily
    schedules: dict = None  # (T, dense) -> TreeSchedule

    def ensure_training(self):
        """The transposed adjacencies and the per-word groupings are only read by the backward pass: a holder
        prepared for inference (they are None then) gets them on its first use under gradient mode."""
        if self.mcsr_t is None:
            h, dev = self.host, self.host["device"]
            with packed_uploads(dev):
                mt_rp, mt_col = transpose_csr(h["m_rp"], h["m_col"], self.n_mess)
                h["mt_rp"], h["mt_col"] = mt_rp, mt_col
                self.mcsr_t = upload_csr(mt_rp, mt_col, dev)
                self.rcsr_t = upload_csr(*transpose_csr(h["r_rp"], h["r_col"], self.n_mess), dev)
                w_rp = np.arange(self.n_words + 1, dtype=np.int32)
                self.word_csr_t = upload_csr(*transpose_csr(w_rp, h["words"].astype(np.int32), self.vocab_size), dev)
                self.key_m_pieces, self.key_m = _key_lists(h["wid_m"], self.n_words, dev)
                self.key_r_pieces, self.key_r = _key_lists(h["wid_root"], self.n_words, dev)
        return self


KEY_CHUNK = 32


def _key_lists(keys: np.ndarray, n_keys: int, device):
    """Inverted index key -> members, as two-level CSR (pieces of <= KEY_CHUNK)."""
    rp, order = K.prep_group_by_key(keys, n_keys)
    piece_rp, key_rp = chunk_lists(rp, KEY_CHUNK)
    pieces = upload_csr(piece_rp, order.astype(np.int64), device)
    key_csr = upload_csr(key_rp, np.arange(piece_rp.shape[0] - 1, dtype=np.int64), device)
    return pieces, key_csr


def _unique_bounded(ids: np.ndarray, bound: int):
    """(sorted distinct values, position of every id among them) for ids in [0, bound): np.unique's result by
    counting instead of sorting."""
    seen = np.bincount(ids, minlength=bound) > 0
    values = np.flatnonzero(seen)
    rank = np.cumsum(seen) - 1
    return values, rank[ids]


def _message_csr(mess_graph, mg: np.ndarray, nm: int):
    """CSR form of a mess_graph table, remembered on the tensor: the deep-tree check and the holder
    preparation both start from it."""
    build = lambda: padded_to_csr(mg, nm)  # noqa: E731
    return cached_prep(mess_graph, ("m_csr", nm), build) if isinstance(mess_graph, torch.Tensor) else build()


def _settling_iterations(mess_graph, m_rp: np.ndarray, m_col: np.ndarray):
    """Uncapped iteration from which every message of a mess_graph table keeps its value, remembered like
    the CSR form (the deep-tree check and the schedule both need it)."""
    build = lambda: K.prep_stable_from(m_rp, m_col, 2 ** 30)  # noqa: E731
    return cached_prep(mess_graph, ("sf_true",), build) if isinstance(mess_graph, torch.Tensor) else build()


def prepare_tree(holder, device, vocab_size: int, train: bool = True) -> TreePrep:
    with packed_uploads(device):
        prep = _prepare_tree(holder, device, vocab_size)
    return prep.ensure_training() if train else prep


def _prepare_tree(holder, device, vocab_size: int) -> TreePrep:
    fnode, fmess, node_graph, mess_graph, scope = holder
    fn, fm = np.asarray(_to_np(fnode), dtype=np.int64), np.asarray(_to_np(fmess), dtype=np.int64)
    ng, mg = _to_np(node_graph), _to_np(mess_graph)
    nn_, nm = fn.shape[0], fm.shape[0]
    if ng.shape[0] != nn_ or mg.shape[0] != nm:
        raise ValueError("node_graph/mess_graph row counts do not match fnode/fmess")
    if nn_ == 0 or nm == 0:
        raise ValueError("empty tree batch")
    check_range(fn, vocab_size, "fnode")
    check_range(fm, nn_, "fmess")
    m_rp, m_col = _message_csr(mess_graph, mg, nm)
    roots = np.asarray([st for st, _ in scope], dtype=np.int64)
    check_range(roots, nn_, "scope")
    r_rp, r_col = padded_to_csr(ng[roots], nm)
    return _tree_prep_from_csr(fn, fm, roots, m_rp, m_col, r_rp, r_col, device, vocab_size,
                               _settling_iterations(mess_graph, m_rp, m_col))


def _tree_prep_from_csr(fn, fm, roots, m_rp, m_col, r_rp, r_col, device, vocab_size: int, sf_true,
                        dev_csr=None) -> TreePrep:
    """fn: word of every node, fm: source node of every message, roots: first node of every tree; m / r: CSR of the
    messages each message reads / that arrive at each root; sf_true: uncapped settling iteration of every
    message.  dev_csr: DeviceCSR forms of m and r when the arrays already sit in upload staging memory."""
    nn_, nm, nb = fn.shape[0], fm.shape[0], roots.shape[0]
    words, wid_node = _unique_bounded(fn, vocab_size)
    nw = words.shape[0]
    wid_m = wid_node[fm]
    wid_root = wid_node[roots]
    w_rp = np.arange(nw + 1, dtype=np.int64)
    ar = np.arange(nb + 1, dtype=np.int64)
    mcsr, rcsr = dev_csr if dev_csr is not None else (upload_csr(m_rp, m_col, device), upload_csr(r_rp, r_col, device))
    # forward-only form; ensure_training() adds the transposes and the per-word groupings
    return TreePrep(
        n_nodes=nn_, n_mess=nm, n_trees=nb, n_words=nw,
        words=_dev_i32(words, device),
        word_csr=upload_csr(w_rp, words, device), word_csr_t=None,
        wid_m=_dev_i32(wid_m, device), wid_root=_dev_i32(wid_root, device),
        mcsr=mcsr, mcsr_t=None,
        rcsr=rcsr, rcsr_t=None,
        ident=upload_csr(ar, ar[:-1], device),
        key_m_pieces=None, key_m=None, key_r_pieces=None, key_r=None,
        vocab_size=vocab_size,
        host=dict(m_rp=m_rp, m_col=m_col, mt_rp=None, mt_col=None, wid_m=wid_m, wid_root=wid_root, words=words,
                  r_rp=r_rp, r_col=r_col, device=device, sf_true=sf_true),
        schedules={},
    )


@dataclass
class DagPrep:
    """Device-side form of a set of GRU messages over an arbitrary dependency graph without cycles
    (the teacher-forced decoder: fast_jtnn/jtnn_dec.py:76-147 creates every directed tree edge once,
    from the messages created before it).  Same fields as the message part of TreePrep."""

    n_mess: int  # including the padding message 0
    n_words: int
    words: torch.Tensor
    word_csr: DeviceCSR
    word_csr_t: DeviceCSR
    wid_m: torch.Tensor
    mcsr: DeviceCSR  # row m: the messages m is computed from, in the reference's neighbour order
    mcsr_t: DeviceCSR
    key_m_pieces: DeviceCSR
    key_m: DeviceCSR
    vocab_size: int
    depth: int  # longest dependency chain: the number of GRU levels
    host: dict = None
    schedules: dict = None


def prepare_message_dag(word_of_mess: np.ndarray, dep_rp: np.ndarray, dep_col: np.ndarray, vocab_size: int,
                        device) -> DagPrep:
    """word_of_mess[m]: word id of the node message m leaves from (entry 0 = padding message, any valid word);
    dep_rp / dep_col: CSR over message ids of the messages each message reads."""
    with packed_uploads(device):
        wm = np.asarray(word_of_mess, dtype=np.int64)
        nm = wm.shape[0]
        check_range(wm, vocab_size, "message words")
        m_rp = np.ascontiguousarray(dep_rp, dtype=np.int32)
        m_col = np.ascontiguousarray(dep_col, dtype=np.int32)
        check_range(m_col, nm, "message dependencies")
        mt_rp, mt_col = transpose_csr(m_rp, m_col, nm)
        words, wid_m = _unique_bounded(wm, vocab_size)
        nw = words.shape[0]
        w_rp = np.arange(nw + 1, dtype=np.int64)
        wt_rp, wt_col = transpose_csr(w_rp.astype(np.int32), words.astype(np.int32), vocab_size)
        km_p, km = _key_lists(wid_m, nw, device)
        depth = int(K.prep_stable_from(m_rp, m_col, 2 ** 30).max()) if nm else 1
        return DagPrep(
            n_mess=nm, n_words=nw, words=_dev_i32(words, device),
            word_csr=upload_csr(w_rp, words, device), word_csr_t=upload_csr(wt_rp, wt_col, device),
            wid_m=_dev_i32(wid_m, device), mcsr=upload_csr(m_rp, m_col, device),
            mcsr_t=upload_csr(mt_rp, mt_col, device), key_m_pieces=km_p, key_m=km, vocab_size=vocab_size,
            depth=depth, host=dict(m_rp=m_rp, m_col=m_col, mt_rp=mt_rp, mt_col=mt_col, wid_m=wid_m, device=device),
            schedules={})


def split_deep_trees(holder, T: int):
    """Trees whose messages settle within T iterations can be computed once per message; a tree with a longer
    chain (its truncated recurrence differs from the fixed point, fast_jtnn/jtnn_enc.py:111) needs the
    iteration-by-iteration schedule with per-iteration state.  Returns None when no tree of the holder is that
    deep, else ((shallow holder, tree ids, message ids), (deep holder, tree ids, message ids)): two holders in
    the reference's layout over disjoint sets of trees, with the positions of their trees and messages in the
    original batch - so that one deep tree does not drag a whole batch onto the slow, memory-hungry schedule."""
    fnode, fmess, node_graph, mess_graph, scope = holder
    mg = np.asarray(_to_np(mess_graph), dtype=np.int64)
    nm = mg.shape[0]
    if nm <= 1 or T < 1:
        return None
    m_rp, m_col = _message_csr(mess_graph, mg, nm)
    sf = _settling_iterations(mess_graph, m_rp, m_col)
    if int(sf.max()) <= T:
        return None
    fn, fm = np.asarray(_to_np(fnode), dtype=np.int64), np.asarray(_to_np(fmess), dtype=np.int64)
    ng = np.asarray(_to_np(node_graph), dtype=np.int64)
    sc = np.asarray(scope, dtype=np.int64).reshape(-1, 2)
    tree_of_node = np.repeat(np.arange(sc.shape[0], dtype=np.int64), sc[:, 1])
    if tree_of_node.shape[0] != fn.shape[0] or not np.array_equal(sc[:, 0], np.concatenate([[0], np.cumsum(sc[:, 1])[:-1]])):
        return None  # not the tensorizer's contiguous layout: leave the batch whole
    tree_of_mess = tree_of_node[fm]
    deep = np.zeros(sc.shape[0], dtype=bool)
    deep[np.unique(tree_of_mess[1:][sf[1:] > T])] = True
    if deep.all():
        return None

    def part(mask):
        trees = np.flatnonzero(mask)
        keep_node = mask[tree_of_node]
        keep_mess = mask[tree_of_mess]
        keep_mess[0] = False
        mess_ids = np.flatnonzero(keep_mess)
        node_new = np.cumsum(keep_node) - 1
        mess_new = np.zeros(nm, dtype=np.int64)
        mess_new[mess_ids] = np.arange(1, mess_ids.shape[0] + 1)
        fm_new = np.concatenate([[0], node_new[fm[mess_ids]]])
        mg_new = np.concatenate([np.zeros((1, mg.shape[1]), dtype=np.int64), mess_new[mg[mess_ids]]], axis=0)
        lens = sc[trees, 1]
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        sub = (torch.from_numpy(fn[keep_node]), torch.from_numpy(fm_new), torch.from_numpy(mess_new[ng[keep_node]]),
               torch.from_numpy(mg_new), [(int(a), int(b)) for a, b in zip(starts, lens)])
        return sub, torch.from_numpy(trees), torch.from_numpy(mess_ids)

    return part(~deep), part(deep)


def tree_schedule(prep: TreePrep, T: int, dense: bool = False, mode: str = None) -> TreeSchedule:
    mode = mode or ("dense" if dense else "once")
    key = (T, mode)
    if key not in prep.schedules:
        h = prep.host
        prep.schedules[key] = build_tree_schedule(h["m_rp"], h["m_col"], h["mt_rp"], h["mt_col"], prep.n_mess, T,
                                                  h["device"], mode=mode, sf_true=h.get("sf_true"))
    return prep.schedules[key]


def backward_schedule(prep: TreePrep, sched: TreeSchedule, all_messages: bool) -> BackwardSchedule:
    """all_messages: the caller's gradient reaches every message (JTMPN reads them); otherwise only the
    messages flowing into a root node carry gradient (tree_vecs is the only consumer)."""
    key = "all" if all_messages else "roots"
    if key not in sched.bwd:
        h = prep.host
        seeds = None if all_messages else h["r_col"]
        sched.bwd[key] = build_backward_schedule(sched, seeds, h["wid_m"], prep.n_words, h["device"])
    return sched.bwd[key]


# ---- active-row schedule of the tree GRU ---------------------------------------------------


@dataclass
class RowList:
    rows: Optional[torch.Tensor]  # int32 [n] message ids, None = all messages in order
    n: int
    nnz: int = 0  # graph entries touched by these rows (for traffic accounting)

    def ptr(self):
        return None if self.rows is None else self.rows.data_ptr()


@dataclass
class BackwardSchedule:
    """Which messages receive gradient at which iteration, and where their rows are stored.

    Walking back from iteration T-1: E[t] are the messages whose h_{t+1} has a non-zero gradient,
    J[t] = the messages they read (= E[t-1]).  Pre-activation gradients of E[t] are stored compactly
    at rows [offE[t], offE[t]+n), the U_r-product gradients of J[t] at [offJ[t], ...).
    """

    E: list  # RowList per iteration t
    J: list  # RowList per iteration t (J[0] unused)
    offE: list
    offJ: list
    nE: int
    nJ: int
    active: Optional[torch.Tensor]  # int64 bitmask per message: bit t set <=> message in E[t]
    # weight-gradient contraction lists (compact gradient row, saved-state row in the slab array)
    zx: torch.Tensor
    zy: torch.Tensor
    n_z: int
    ux: torch.Tensor
    uy: torch.Tensor
    n_u: int
    key_pieces: DeviceCSR  # compact E rows grouped by word-table row
    key: DeviceCSR
    once: bool = False  # E / J are reverse-dependency levels (each message once), not iterations
    n_dep: list = None  # once: leading rows of E[L] that read other messages (the rest read nothing)


@dataclass
class TreeSchedule:
    T: int
    dense: bool
    sf: torch.Tensor  # int32 [NM] iteration from which a message no longer changes (capped at T)
    sel_final: torch.Tensor  # int32 [NM] slab that holds the final value: min(T, sf)
    F: list  # RowList per iteration: messages to update (level: sf > t; once: sf == t + 1)
    U: list  # RowList per iteration t >= 1: messages whose U_r product is needed (changed at t-1, have readers)
    bwd: dict  # "all" / "roots" -> BackwardSchedule, built on first use
    _host: dict = None
    mode: str = "level"  # "dense", "level" or "once" (see build_tree_schedule)


def _levels(rowptr, col, n, cap, start):
    """level[m] = start if row m has no entries else 1 + max(level[col entries]), capped."""
    deg = np.diff(rowptr)
    row_of = np.repeat(np.arange(n, dtype=np.int64), deg)
    lev = np.full(n, start, dtype=np.int64)
    for _ in range(cap + 1):
        mx = np.full(n, start - 1, dtype=np.int64)
        if col.size:
            np.maximum.at(mx, row_of, lev[col])
        new = np.minimum(np.where(deg > 0, mx + 1, start), cap)
        if np.array_equal(new, lev):
            break
        lev = new
    return lev


def build_tree_schedule(m_rp, m_col, mt_rp, mt_col, nm, T, device, dense=False, mode: str = None,
                        sf_true=None) -> TreeSchedule:
    """Row lists of the T GraphGRU iterations.

    mode "dense": every message in every iteration, the reference's schedule (jtnn_enc.py:107-128).
    mode "level": iteration t updates the messages that can still change (stable_from > t).
    mode "once":  iteration t updates only the messages that BECOME stable in it (stable_from == t + 1).
        A message's final value is f(final values of the messages it reads); the values it takes before
        its inputs have settled are read by nobody once the recurrence has run to its fixed point, so when
        every message settles within T iterations (tree depth <= T) computing each message once, level by
        level, gives the reference's result.  Batches with a deeper tree fall back to "level", where the
        truncated recurrence is followed iteration by iteration.
    """
    mode = mode or ("dense" if dense else "once")
    if mode not in ("dense", "level", "once"):
        raise ValueError("unknown schedule mode %r" % (mode,))
    if nm > 0 and mode != "dense":
        if sf_true is None:  # uncapped; a caller that has it already (the deep-tree check) hands it in
            sf_true = K.prep_stable_from(m_rp, m_col, 2 ** 30)
    else:
        sf_true = None
    if mode == "once" and (T < 1 or (nm > 0 and int(sf_true.max()) > T)):
        mode = "level"  # a tree deeper than the iteration count: follow the truncated recurrence
    with packed_uploads(device):
        return _build_tree_schedule(m_rp, m_col, mt_rp, mt_col, nm, T, device, mode, sf_true)


def _build_tree_schedule(m_rp, m_col, mt_rp, mt_col, nm, T, device, mode, sf_true=None) -> TreeSchedule:
    dense = mode == "dense"
    once = mode == "once"
    deg = np.diff(m_rp)
    F, U = [], []
    total_nnz = int(m_col.shape[0])
    if once:
        # messages grouped by the iteration in which they settle, message order inside a group: iteration t
        # updates group t + 1 and needs the U_r products of group t's members that somebody reads (one native pass)
        sf, order, keyptr, u_rows, u_ptr, nnz_ptr = K.prep_forward_once(m_rp, m_col, sf_true, T)
        order_dev = _dev_i32(order, device) if nm else None
        u_dev = _dev_i32(u_rows, device) if u_rows.size else None
        tdeg = None
    else:
        # readers per message: from the transposed lists when they exist, else counted (inference-only holders)
        tdeg = np.diff(mt_rp) if mt_rp is not None else np.bincount(m_col, minlength=nm)
        if dense or T < 1 or sf_true is None:
            sf = np.full(nm, max(T, 1), dtype=np.int64)
        else:
            sf = np.minimum(sf_true.astype(np.int64), max(T, 1))
    for t in range(T):
        if dense or t == 0:
            F.append(RowList(None, nm, total_nnz))
        elif once:
            a, b = int(keyptr[t + 1]), int(keyptr[t + 2])
            F.append(RowList(order_dev[a:b] if b > a else None, b - a, int(nnz_ptr[t + 2] - nnz_ptr[t + 1])))
        else:
            r = np.flatnonzero(sf > t)
            F.append(RowList(_dev_i32(r, device) if r.size else None, int(r.size), int(deg[r].sum())))
        if t == 0:
            U.append(RowList(None, 0))
        elif dense:
            U.append(RowList(None, nm))
        elif once:
            a, b = int(u_ptr[t]), int(u_ptr[t + 1])
            U.append(RowList(u_dev[a:b] if b > a else None, b - a))
        else:
            # messages whose U_r product is read in this iteration and not yet available: those that took
            # their current value in iteration t-1 and have readers
            r = np.flatnonzero((sf >= t) & (tdeg > 0))
            U.append(RowList(_dev_i32(r, device) if r.size else None, int(r.size)))
    # the CSR arrays stay as they came (int32): the backward plans widen what they index with, when they need to
    host = dict(m_rp=m_rp, m_col=m_col, mt_rp=mt_rp, mt_col=mt_col, sf=sf, nm=nm, deg=deg.astype(np.int64),
                tdeg=None if tdeg is None else tdeg.astype(np.int64))
    return TreeSchedule(T=T, dense=dense, sf=_dev_i32(sf, device), sel_final=_dev_i32(np.minimum(sf, T), device),
                        F=F, U=U, bwd={}, _host=host, mode=mode)


def build_backward_schedule(sched: TreeSchedule, seeds: Optional[np.ndarray], wid_m: np.ndarray, n_words: int,
                            device) -> BackwardSchedule:
    """seeds: message ids with a non-zero external gradient on the final h (None = all)."""
    with packed_uploads(device):
        return _build_backward_schedule(sched, seeds, wid_m, n_words, device)


def _expand_lists(rp: np.ndarray, col: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """Concatenation of the CSR lists of ``rows``."""
    starts, cnt = rp[rows], rp[rows + 1] - rp[rows]
    tot = int(cnt.sum())
    if tot == 0:
        return col[:0]
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(tot)
    return col[idx]


def backward_once_plan_np(m_rp, m_col, mt_rp, mt_col, deg, nm, seeds):
    """numpy statement of K.prep_backward_once (its checker)."""
    act = np.zeros(nm, dtype=bool)
    if seeds is None:
        act[:] = True
    else:
        cur = np.unique(seeds.astype(np.int64))
        act[cur] = True
        while cur.size:  # closure over "is read by a gradient-carrying message"
            nxt = np.unique(_expand_lists(m_rp, m_col, cur))
            cur = nxt[~act[nxt]]
            act[cur] = True
    # reverse level = 1 + max over the active readers (0 without one): relaxation over the transposed graph
    # restricted to active readers
    if seeds is None:
        f_rp, f_col = mt_rp, mt_col
    else:
        keep = act[mt_col]
        row_of = np.repeat(np.arange(nm, dtype=np.int64), np.diff(mt_rp))
        f_col = mt_col[keep]
        f_rp = np.zeros(nm + 1, dtype=np.int64)
        np.cumsum(np.bincount(row_of[keep], minlength=nm), out=f_rp[1:])
    rl = K.prep_stable_from(f_rp, f_col, 2 ** 30).astype(np.int64) - 1
    n_lev = int(rl[act].max()) + 1 if act.any() else 0
    order = np.flatnonzero(act)
    # within a level: rows with inputs first, message order otherwise
    has_dep = deg[order] > 0
    order = order[np.lexsort((order, ~has_dep, rl[order]))]
    lev_cnt = np.bincount(rl[order], minlength=max(n_lev, 1))[:n_lev]
    lev_off = np.concatenate([[0], np.cumsum(lev_cnt)]).astype(np.int64)
    tdeg_act = np.diff(f_rp)
    lv = [order[lev_off[L]:lev_off[L + 1]] for L in range(n_lev)]
    return dict(act=act, order=order, lev_off=lev_off, n_dep=np.asarray([int((deg[r] > 0).sum()) for r in lv]),
                deg_sum=np.asarray([int(deg[r].sum()) for r in lv]),
                tdeg_sum=np.asarray([int(tdeg_act[r].sum()) for r in lv]))


def _dev_i64(a: np.ndarray, device) -> torch.Tensor:
    a = np.ascontiguousarray(a, dtype=np.int64)
    if _TLS.upload is not None and _TLS.upload.device == torch.device(device) and a.size:
        return _TLS.upload.add(a.view(np.int32)).view(torch.int64)  # chunks are 16-byte aligned
    return _new_device_tensor(torch.from_numpy(a).to(device))


def _build_backward_schedule_once(sched: TreeSchedule, seeds: Optional[np.ndarray], wid_m: np.ndarray,
                                  n_words: int, device) -> BackwardSchedule:
    """Backward pass of the compute-once schedule: every message that carries gradient is visited once.

    A message's total gradient is what its consumers outside the recurrence hand in plus what the
    messages that read it pass back, so it is complete once all of those readers have been visited:
    level 0 holds the messages no gradient-carrying message reads, level L those whose readers all sit
    below L.  E[L] lists level L (rows that read other messages first: only they have products to
    propagate), J[L] = E[L] for L >= 1 are the rows whose incoming gradient is gathered first.
    """
    h = sched._host
    deg = h["deg"]
    plan = K.prep_backward_once(h["m_rp"], h["m_col"], seeds)
    order = plan["order"]  # int32: uploaded as it is
    lev_off, n_dep = plan["lev_off"], [int(v) for v in plan["n_dep"]]
    n_lev = len(n_dep)
    nE = int(order.size)
    order_dev = _dev_i32(order, device) if nE else None  # one array; the levels are slices of it
    El, Jl, offE, offJ = [], [], [], []
    nJ = 0
    for L in range(n_lev):
        a, b = int(lev_off[L]), int(lev_off[L + 1])
        offE.append(a)
        rows_t = order_dev[a:b]
        El.append(RowList(rows_t, b - a, int(plan["deg_sum"][L])))
        if L == 0:
            Jl.append(RowList(None, 0))
            offJ.append(0)
        else:
            Jl.append(RowList(rows_t, b - a, int(plan["tdeg_sum"][L])))
            offJ.append(nJ)
            nJ += b - a
    dep_rows = deg[order] > 0  # rows with s, g, t state: they contribute to the W_z / W_h products
    zx, zy = np.flatnonzero(dep_rows).astype(np.int32), order[dep_rows]  # positions < nE <= nm: int32 holds them
    lev0 = int(lev_off[1]) if n_lev else 0
    ux, uy = np.arange(nJ, dtype=np.int32), order[lev0:]
    kp, kc = _key_lists(wid_m[order], n_words, device)
    one = np.zeros(1, dtype=np.int32)
    return BackwardSchedule(
        E=El, J=Jl, offE=offE, offJ=offJ, nE=nE, nJ=nJ,
        active=_dev_i64(plan["act"].astype(np.int64), device),
        zx=_dev_i32(zx if zx.size else one, device), zy=_dev_i32(zy if zy.size else one, device), n_z=int(zx.size),
        ux=_dev_i32(ux if ux.size else one, device), uy=_dev_i32(uy if uy.size else one, device), n_u=int(ux.size),
        key_pieces=kp, key=kc, once=True, n_dep=n_dep)


def _build_backward_schedule(sched: TreeSchedule, seeds: Optional[np.ndarray], wid_m: np.ndarray, n_words: int,
                             device) -> BackwardSchedule:
    if sched.mode == "once":
        return _build_backward_schedule_once(sched, seeds, wid_m, n_words, device)
    h = sched._host
    nm, T, sf = h["nm"], sched.T, h["sf"]
    m_rp, m_col, deg = np.asarray(h["m_rp"], dtype=np.int64), np.asarray(h["m_col"], dtype=np.int64), h["deg"]
    E = [None] * T
    use_mask = (not sched.dense) and T <= 63
    if sched.dense or not use_mask:
        sets = [np.arange(nm, dtype=np.int64)] * T
    else:
        cur = np.arange(nm, dtype=np.int64) if seeds is None else np.unique(seeds.astype(np.int64))
        sets = [None] * T
        for t in range(T - 1, -1, -1):
            sets[t] = cur
            if cur.size == 0:
                for u in range(t - 1, -1, -1):
                    sets[u] = cur
                break
            # messages read by the current set
            starts, cnt = m_rp[cur], deg[cur]
            if int(cnt.sum()) == 0:
                cur = cur[:0]
            else:
                idx = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
                cur = np.unique(m_col[idx])
    offE, offJ = [0] * T, [0] * T
    nE = nJ = 0
    active = np.zeros(nm, dtype=np.int64)
    El, Jl = [], []
    tdeg = h["tdeg"]
    for t in range(T):
        r = sets[t]
        offE[t] = nE
        nE += int(r.size)
        full = r.size == nm
        El.append(RowList(None if full else (_dev_i32(r, device) if r.size else None), int(r.size)))
        if use_mask and r.size:
            active[r] |= np.int64(1) << np.int64(t)
    for t in range(T):
        if t == 0:
            Jl.append(RowList(None, 0))
            continue
        r = sets[t - 1]
        offJ[t] = nJ
        nJ += int(r.size)
        full = r.size == nm
        Jl.append(RowList(None if full else (_dev_i32(r, device) if r.size else None), int(r.size),
                          int(tdeg[r].sum())))
    # weight-gradient contraction lists; saved-state rows are addressed as slab * slab_rows + message, and the
    # compute-once schedule keeps a single slab
    slab_rows = 0 if sched.mode == "once" else nm
    zx, zy, ux, uy, keys = [], [], [], [], []
    for t in range(T):
        r = sets[t]
        if r.size == 0:
            continue
        keys.append(wid_m[r])
        if t >= 1:
            tau = np.minimum(t, sf[r] - 1)
            ok = tau >= 1  # iteration 0 has s == g == 0: no contribution
            zx.append(offE[t] + np.flatnonzero(ok))
            zy.append(tau[ok] * slab_rows + r[ok])
            j = sets[t - 1]
            if j.size:
                ux.append(offJ[t] + np.arange(j.size))
                uy.append(np.minimum(t, sf[j]) * slab_rows + j)
    cat = lambda l: np.concatenate(l) if l else np.zeros(0, dtype=np.int64)  # noqa: E731
    zx, zy, ux, uy, keys = cat(zx), cat(zy), cat(ux), cat(uy), cat(keys)
    if (T + 1) * nm >= 2 ** 31:
        raise OverflowError("slab row index exceeds int32")
    kp, kc = _key_lists(keys.astype(np.int64), n_words, device)
    one = np.zeros(1, dtype=np.int64)
    return BackwardSchedule(
        E=El, J=Jl, offE=offE, offJ=offJ, nE=nE, nJ=nJ,
        active=_new_device_tensor(torch.from_numpy(active).to(device)) if use_mask else None,
        zx=_dev_i32(zx if zx.size else one, device), zy=_dev_i32(zy if zy.size else one, device), n_z=int(zx.size),
        ux=_dev_i32(ux if ux.size else one, device), uy=_dev_i32(uy if uy.size else one, device), n_u=int(ux.size),
        key_pieces=kp, key=kc)


# ---- caching on the holder object --------------------------------------------------------

_CACHE_ATTR = "_jtvae_prep"


class PreparedHolder(tuple):
    """A holder tuple that remembers its device-side form.

    ``tuple`` subclass, so ``model(*holder)`` and unpacking work as with the
    reference's plain tuples.
    """

    def __new__(cls, items):
        self = super().__new__(cls, items)
        self._prep = {}
        return self


def cached_prep(key_tensor, tag, build):
    """Cache by identity of the (CPU) index tensor the caller passes in every step."""
    cache = getattr(key_tensor, _CACHE_ATTR, None)
    if cache is None:
        cache = {}
        try:
            setattr(key_tensor, _CACHE_ATTR, cache)
        except AttributeError:
            return build()
    if tag not in cache:
        cache[tag] = build()
    return cache[tag]
