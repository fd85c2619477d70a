"""Index builders: abstract graphs -> the reference's holder tuples.

Output layout is the reference's, bit for bit (checked against the tables its own tensorizers produce,
tests/golden): 1-based directed bond ids with a zero padding row for the molecular graphs
(fast_jtnn/mpn.py:112-161), message ids in node/neighbour order with trailing zero padding for the trees
(fast_jtnn/jtnn_enc.py:54-92), bond ids offset by the number of tree messages for the candidate graphs
(fast_jtnn/jtmpn.py:113-183).

``mpn_tensorize``, ``jtenc_tensorize(_nodes)``, ``jtmpn_tensorize`` and ``tensorize`` flatten the objects they
are given (``flat.py``) and call the native builders of ``libjtvae_hip.so``; the ``*_np`` functions are the same
tables stated in numpy, kept as checkers of the native ones and for inputs the flat form does not hold
(feature values that are not whole numbers).

Inputs are feature arrays and bond lists; producing those from SMILES is RDKit work that stays with the caller.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .synthetic import JTMPN_MAX_NB, MPN_MAX_NB


def _ragged_to_padded(owner: np.ndarray, value: np.ndarray, n_rows: int, width: int, what: str) -> np.ndarray:
    """Scatter ``value`` into row ``owner`` of a zero table, filling slots left to right in input order."""
    out = np.zeros((n_rows, width), dtype=np.int64)
    if owner.size == 0:
        return out
    order = np.argsort(owner, kind="stable")
    o, v = owner[order], value[order]
    first = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(o, minlength=n_rows), out=first[1:])
    slot = np.arange(o.shape[0], dtype=np.int64) - first[o]
    if slot.size and int(slot.max()) >= width:
        raise IndexError("%s: a row has %d neighbours, table width is %d" % (what, int(slot.max()) + 1, width))
    out[o, slot] = v
    return out


def mpn_tensorize_np(mols: Sequence):
    """numpy statement of mpn_tensorize."""
    n_atoms = np.asarray([m.atom_feats.shape[0] for m in mols], dtype=np.int64)
    n_bonds = np.asarray([m.bonds.shape[0] for m in mols], dtype=np.int64)
    atom_off = np.concatenate([[0], np.cumsum(n_atoms)])
    total_atoms = int(atom_off[-1])
    scope = [(int(atom_off[i]), int(n_atoms[i])) for i in range(len(mols))]
    fatoms = np.concatenate([m.atom_feats for m in mols], axis=0).astype(np.float32)
    bfeat = np.concatenate([m.bond_feats for m in mols], axis=0).astype(np.float32)
    bonds = np.concatenate([m.bonds + atom_off[i] for i, m in enumerate(mols)], axis=0).astype(np.int64)
    nb = bonds.shape[0]
    # directed bond 2i+1 = (begin -> end), 2i+2 = (end -> begin); id 0 is padding
    src = np.empty(2 * nb + 1, dtype=np.int64)
    dst = np.empty(2 * nb + 1, dtype=np.int64)
    src[0] = dst[0] = -1
    src[1::2], dst[1::2] = bonds[:, 0], bonds[:, 1]
    src[2::2], dst[2::2] = bonds[:, 1], bonds[:, 0]
    fbonds = np.zeros((2 * nb + 1, fatoms.shape[1] + bfeat.shape[1]), dtype=np.float32)
    fbonds[1:, :fatoms.shape[1]] = fatoms[src[1:]]
    fbonds[1::2, fatoms.shape[1]:] = bfeat
    fbonds[2::2, fatoms.shape[1]:] = bfeat
    ids = np.arange(1, 2 * nb + 1, dtype=np.int64)
    # in_bonds[a] = directed bonds that END in a, in creation order
    agraph = _ragged_to_padded(dst[1:], ids, total_atoms, MPN_MAX_NB, "agraph")
    # bgraph[b1, i] = in_bonds[x][i] unless it is the reverse bond (comes from y); slot i kept
    bgraph = np.zeros((2 * nb + 1, MPN_MAX_NB), dtype=np.int64)
    if nb:
        cand = agraph[src[1:]]  # [2nb, 6] incoming bonds of the start atom
        cand_src = np.where(cand > 0, src[np.maximum(cand, 0)], -2)
        keep = (cand > 0) & (cand_src != dst[1:, None])
        bgraph[1:] = np.where(keep, cand, 0)
    return (torch.from_numpy(fatoms), torch.from_numpy(fbonds), torch.from_numpy(agraph),
            torch.from_numpy(bgraph), scope)


def jtenc_tensorize_nodes_np(node_batch: Sequence, scope):
    """numpy statement of jtenc_tensorize_nodes."""
    fnode = np.asarray([x.wid for x in node_batch], dtype=np.int64)
    xs, ys = [], []
    for x in node_batch:
        xi = x.idx
        for y in x.neighbors:
            xs.append(xi)
            ys.append(y.idx)
    xs = np.asarray(xs, dtype=np.int64)
    ys = np.asarray(ys, dtype=np.int64)
    nm = xs.shape[0] + 1
    mid = np.arange(1, nm, dtype=np.int64)
    mess_dict = {(int(a), int(b)): int(i) for a, b, i in zip(xs, ys, mid)}
    fmess = np.zeros(nm, dtype=np.int64)
    fmess[1:] = xs
    n_nodes = len(node_batch)
    # node_graph[y] lists messages x->y in message order
    deg_in = np.bincount(ys, minlength=n_nodes) if ys.size else np.zeros(n_nodes, dtype=np.int64)
    kn = max(int(deg_in.max()) if n_nodes else 0, 1)
    node_graph = _ragged_to_padded(ys, mid, n_nodes, kn, "node_graph")
    # mess_graph[(y->z)] lists messages (x->y), x != z, in message order of (x->y):
    # pair every message m1=(x->y) with the outgoing messages of y
    if xs.size:
        # outgoing messages of a node are consecutive ids, numbered node by node in batch order:
        # map a node's idx to the id of its first outgoing message
        order_nodes = np.asarray([x.idx for x in node_batch], dtype=np.int64)
        out_cnt = np.asarray([len(x.neighbors) for x in node_batch], dtype=np.int64)
        start_by_pos = np.concatenate([[0], np.cumsum(out_cnt)[:-1]]) + 1
        start = np.zeros(n_nodes, dtype=np.int64)
        cnt = np.zeros(n_nodes, dtype=np.int64)
        start[order_nodes] = start_by_pos
        cnt[order_nodes] = out_cnt
        rep = cnt[ys]  # for each m1, number of outgoing messages of its target y
        m1 = np.repeat(mid, rep)
        base = np.repeat(start[ys], rep)
        within = np.arange(m1.shape[0], dtype=np.int64) - np.repeat(np.concatenate([[0], np.cumsum(rep)[:-1]]), rep)
        m2 = base + within
        keep = ys[m2 - 1] != xs[m1 - 1]  # drop z == x
        m1, m2 = m1[keep], m2[keep]
        deg_m = np.bincount(m2, minlength=nm)
        km = max(int(deg_m.max()), 1)
        mess_graph = _ragged_to_padded(m2, m1, nm, km, "mess_graph")
    else:
        mess_graph = np.zeros((nm, 1), dtype=np.int64)
    return (torch.from_numpy(fnode), torch.from_numpy(fmess), torch.from_numpy(node_graph),
            torch.from_numpy(mess_graph), scope), mess_dict


def jtmpn_tensorize_np(cand_batch: Sequence, mess_dict):
    """numpy statement of jtmpn_tensorize."""
    total_mess = len(mess_dict) + 1
    n_atoms = np.asarray([c.atom_feats.shape[0] for c in cand_batch], dtype=np.int64)
    atom_off = np.concatenate([[0], np.cumsum(n_atoms)])
    total_atoms = int(atom_off[-1])
    scope = [(int(atom_off[i]), int(n_atoms[i])) for i in range(len(cand_batch))]
    fatoms = np.concatenate([c.atom_feats for c in cand_batch], axis=0).astype(np.float32)
    bfeat = np.concatenate([c.bond_feats for c in cand_batch], axis=0).astype(np.float32)
    bonds = np.concatenate([c.bonds + atom_off[i] for i, c in enumerate(cand_batch)], axis=0).astype(np.int64)
    # batch-global tree node id of every atom (-1: unlabeled), jtmpn.py:140-142
    bid = np.concatenate([
        np.where(c.atom_nid > 0,
                 np.asarray([n.idx for n in c.all_nodes], dtype=np.int64)[np.maximum(c.atom_nid - 1, 0)], -1)
        for c in cand_batch])
    nb = bonds.shape[0]
    src = np.empty(2 * nb, dtype=np.int64)
    dst = np.empty(2 * nb, dtype=np.int64)
    src[0::2], dst[0::2] = bonds[:, 0], bonds[:, 1]
    src[1::2], dst[1::2] = bonds[:, 1], bonds[:, 0]
    fbonds = np.zeros((2 * nb, fatoms.shape[1] + bfeat.shape[1]), dtype=np.float32)
    fbonds[:, :fatoms.shape[1]] = fatoms[src]
    fbonds[0::2, fatoms.shape[1]:] = bfeat
    fbonds[1::2, fatoms.shape[1]:] = bfeat
    # in_bonds entries in creation order.  Per undirected bond i (x,y): bond 2i ends in y, bond 2i+1
    # ends in x, then tree message (x_bid,y_bid) ends in y, then (y_bid,x_bid) ends in x (jtmpn.py:146-162)
    xb, yb = bid[bonds[:, 0]], bid[bonds[:, 1]]
    labeled = (xb >= 0) & (yb >= 0) & (xb != yb)
    m_xy = np.asarray([mess_dict.get((int(a), int(b)), 0) if l else 0 for a, b, l in zip(xb, yb, labeled)],
                      dtype=np.int64)
    m_yx = np.asarray([mess_dict.get((int(b), int(a)), 0) if l else 0 for a, b, l in zip(xb, yb, labeled)],
                      dtype=np.int64)
    owner = np.stack([bonds[:, 1], bonds[:, 0], bonds[:, 1], bonds[:, 0]], axis=1).reshape(-1)
    value = np.stack([total_mess + 2 * np.arange(nb), total_mess + 2 * np.arange(nb) + 1, m_xy, m_yx],
                     axis=1).reshape(-1).astype(np.int64)
    valid = np.stack([np.ones(nb, bool), np.ones(nb, bool), m_xy > 0, m_yx > 0], axis=1).reshape(-1)
    agraph = _ragged_to_padded(owner[valid], value[valid], total_atoms, JTMPN_MAX_NB, "agraph")
    bgraph = np.zeros((2 * nb, JTMPN_MAX_NB), dtype=np.int64)
    if nb:
        cand = agraph[src]  # in_bonds of the start atom x of every directed bond
        is_bond = cand >= total_mess
        cand_src = np.where(is_bond, src[np.clip(cand - total_mess, 0, 2 * nb - 1)], -2)
        keep = (cand > 0) & (~is_bond | (cand_src != dst[:, None]))
        bgraph = np.where(keep, cand, 0)
    return (torch.from_numpy(fatoms), torch.from_numpy(fbonds), torch.from_numpy(agraph),
            torch.from_numpy(bgraph), scope)


# ---- the native builders behind the reference's names ---------------------------------------------


def _scope_list(scope: np.ndarray):
    return [(int(a), int(b)) for a, b in scope.tolist()]


def _graph_holder(tables):
    fa, fb, ag, bg, scope = tables
    return (torch.from_numpy(fa), torch.from_numpy(fb), torch.from_numpy(ag), torch.from_numpy(bg), _scope_list(scope))


def mpn_tensorize(mols: Sequence):
    """(fatoms, fbonds, agraph, bgraph, scope) as fast_jtnn/mpn.py:110-161 builds them."""
    from . import flat

    try:
        gs = flat.graph_set_from_objects(mols)
    except ValueError:  # feature values the byte form cannot hold
        return mpn_tensorize_np(mols)
    return _graph_holder(flat.graph_tables(gs, max_nb=MPN_MAX_NB))


def jtenc_tensorize(tree_batch: Sequence):
    """fast_jtnn/jtnn_enc.py:44-52."""
    node_batch: List = []
    scope = []
    for tree in tree_batch:
        scope.append((len(node_batch), len(tree.nodes)))
        node_batch.extend(tree.nodes)
    return jtenc_tensorize_nodes(node_batch, scope)


def jtenc_tensorize_nodes(node_batch: Sequence, scope):
    """((fnode, fmess, node_graph, mess_graph, scope), mess_dict) as fast_jtnn/jtnn_enc.py:54-92.  ``mess_dict``
    is a read-only mapping over the message arrays (``flat.MessDict``)."""
    from . import flat

    sizes = [n for _, n in scope]
    starts = np.asarray([st for st, _ in scope], dtype=np.int64)
    contiguous = sum(sizes) == len(node_batch) and np.array_equal(np.cumsum([0] + sizes[:-1]), starts)
    try:
        if not contiguous:
            raise ValueError("scope is not the trees back to back")
        ts, _ = flat.tree_set_from_nodes(node_batch, sizes)
    except ValueError:  # node ids that are not batch positions, neighbours outside their tree's scope ...
        return jtenc_tensorize_nodes_np(node_batch, scope)
    try:
        (fnode, fmess, node_graph, mess_graph, _), (src, dst) = flat.tree_tables(ts)
    except IndexError:  # a neighbour outside its own tree: the scope does not delimit the trees
        return jtenc_tensorize_nodes_np(node_batch, scope)
    holder = (torch.from_numpy(fnode), torch.from_numpy(fmess), torch.from_numpy(node_graph),
              torch.from_numpy(mess_graph), scope)
    return holder, flat.MessDict(src, dst, len(node_batch))


def jtmpn_tensorize(cand_batch: Sequence, mess_dict):
    """(fatoms, fbonds, agraph, bgraph, scope) as fast_jtnn/jtmpn.py:113-183."""
    from . import flat

    if not len(cand_batch):
        return _empty_cand_holder()
    try:
        gs = flat.graph_set_from_objects(cand_batch, with_nid=True)
        src, dst, n_nodes = flat.mess_arrays(mess_dict)
    except ValueError:
        return jtmpn_tensorize_np(cand_batch, mess_dict)
    # atom-map number k of a candidate names node all_nodes[k - 1]; with batch-wide consecutive node ids that
    # is all_nodes[0].idx + k - 1
    base = np.empty(len(cand_batch), np.int64)
    seen = {}
    for i, c in enumerate(cand_batch):
        key = id(c.all_nodes)
        b = seen.get(key)
        if b is None:
            nodes = c.all_nodes
            b = nodes[0].idx if len(nodes) else 0
            if any(n.idx != b + j for j, n in enumerate(nodes)):
                return jtmpn_tensorize_np(cand_batch, mess_dict)
            seen[key] = b
            n_nodes = max(n_nodes, b + len(nodes))
        base[i] = b
    mess = flat.candidate_mess(src, dst, n_nodes, base)
    return _graph_holder(flat.graph_tables(gs, mess=mess, max_nb=JTMPN_MAX_NB))


def _empty_cand_holder():
    from .synthetic import JTMPN_ATOM_FDIM, JTMPN_BOND_FDIM

    return (torch.zeros((0, JTMPN_ATOM_FDIM)), torch.zeros((0, JTMPN_ATOM_FDIM + JTMPN_BOND_FDIM)),
            torch.zeros((0, JTMPN_MAX_NB), dtype=torch.int64), torch.zeros((0, JTMPN_MAX_NB), dtype=torch.int64), [])


def tensorize(tree_batch: Sequence, vocab=None, assm: bool = True):
    """The per-batch ``tensorize`` of fast_jtnn/datautils.py:140-163 on abstract trees (``tree.mol`` holds the
    molecule's graph, ``node.cands`` the candidate graphs): returns ``(tree_batch, jtenc_holder, mpn_holder)``
    and, with ``assm``, ``(jtmpn_holder, batch_idx)`` as a fourth item."""
    from .synthetic import collect_candidates, set_batch_nodeID

    set_batch_nodeID(tree_batch, vocab)
    jtenc_holder, mess_dict = jtenc_tensorize(tree_batch)
    mpn_holder = mpn_tensorize([t.mol for t in tree_batch])
    if assm is False:
        return tree_batch, jtenc_holder, mpn_holder
    cands, batch_idx = collect_candidates(tree_batch)
    return tree_batch, jtenc_holder, mpn_holder, (jtmpn_tensorize(cands, mess_dict), torch.from_numpy(batch_idx))
