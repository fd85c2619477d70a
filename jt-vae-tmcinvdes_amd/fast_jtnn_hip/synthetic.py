"""Seed-driven synthetic molecules, junction trees and attachment candidates.

RDKit is what turns SMILES into graphs in the reference (fast_jtnn/mpn.py:110-161,
fast_jtnn/mol_tree.py:78-121, fast_jtnn/jtmpn.py:113-183).  It is not needed for
the encoder hot path itself, which only sees feature matrices and index tensors.
This module produces *abstract* graphs with the same feature widths and the same
degree limits, so that the tensorizers and the encoder kernels can be exercised
and benchmarked without chemistry.

All randomness comes from ``numpy.random.RandomState`` (a frozen stream), so a
seed names one batch exactly, independent of the torch RNG.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# Feature widths of the molecular-graph encoder (fast_jtnn/mpn.py:35-37).
MPN_ELEMS = 23
MPN_ATOM_FDIM = MPN_ELEMS + 6 + 5 + 4 + 1  # 39
MPN_BOND_FDIM = 6 + 6  # 12 (5 bond types + in-ring, 6 stereo classes)
MPN_MAX_NB = 6
# Feature widths of the candidate-graph encoder (fast_jtnn/jtmpn.py:37-39).
JTMPN_ELEMS = 26
JTMPN_ATOM_FDIM = JTMPN_ELEMS + 6 + 5 + 1  # 38
JTMPN_BOND_FDIM = 6
JTMPN_MAX_NB = 15


@dataclass
class MolGraph:
    """One molecule as the MPN tensorizer sees it: atoms, undirected bonds, features."""

    atom_feats: np.ndarray  # [n_atoms, 39] float32, one-hot blocks
    bonds: np.ndarray  # [n_bonds, 2] int64 (begin atom, end atom), molecule-local
    bond_feats: np.ndarray  # [n_bonds, 12] float32

    @property
    def n_atoms(self) -> int:
        return int(self.atom_feats.shape[0])


@dataclass(eq=False)  # candidates compare by identity, like the distinct SMILES strings they stand in for
class CandGraph:
    """One attachment candidate as the JTMPN tensorizer sees it.

    ``atom_nid[a]`` plays the role of RDKit's atom-map number
    (fast_jtnn/jtmpn.py:140): 0 = unlabeled, otherwise a 1-based position in
    ``all_nodes`` of the tree node the atom belongs to.
    """

    atom_feats: np.ndarray  # [n_atoms, 38] float32
    bonds: np.ndarray  # [n_bonds, 2] int64
    bond_feats: np.ndarray  # [n_bonds, 6] float32
    atom_nid: np.ndarray  # [n_atoms] int64
    all_nodes: list  # nodes of the owning tree (objects with .idx)
    ctr_node: object

    @property
    def n_atoms(self) -> int:
        return int(self.atom_feats.shape[0])


class TreeNode:
    """Junction-tree node with the attributes the encoder tensorizer reads
    (fast_jtnn/jtnn_enc.py:54-76, fast_jtnn/datautils.py:166-172)."""

    __slots__ = ("nid", "idx", "wid", "neighbors", "is_leaf", "cands", "label", "smiles")

    def __init__(self, nid: int, wid: int):
        self.nid = nid  # 1-based position inside its tree
        self.idx = -1  # batch-global index, set by set_batch_nodeID
        self.wid = wid
        self.neighbors: List["TreeNode"] = []
        self.is_leaf = False
        self.cands: List[CandGraph] = []
        self.label: Optional[CandGraph] = None
        self.smiles = "w%d" % wid


@dataclass
class Tree:
    nodes: List[TreeNode]
    mol: MolGraph
    smiles: str = ""


class FakeVocab:
    """Vocabulary stand-in: node "smiles" are ``w<id>`` strings."""

    def __init__(self, size: int):
        self._size = size

    def size(self) -> int:
        return self._size

    def get_index(self, smiles: str) -> int:
        return int(smiles[1:])


def set_batch_nodeID(tree_batch: Sequence[Tree], vocab=None) -> None:
    """Number nodes consecutively over the batch (fast_jtnn/datautils.py:166-172)."""
    tot = 0
    for tree in tree_batch:
        for node in tree.nodes:
            node.idx = tot
            if vocab is not None:
                node.wid = vocab.get_index(node.smiles)
            tot += 1


def _one_hot_blocks(rs: np.random.RandomState, n: int, blocks: Sequence[int], tail_bits: int) -> np.ndarray:
    width = sum(blocks) + tail_bits
    out = np.zeros((n, width), dtype=np.float32)
    off = 0
    rows = np.arange(n)
    for b in blocks:
        out[rows, off + rs.randint(0, b, size=n)] = 1.0
        off += b
    if tail_bits:
        out[:, off:] = (rs.rand(n, tail_bits) < 0.35).astype(np.float32)
    return out


def _random_bonds(rs: np.random.RandomState, n: int, tree_deg_cap: int, max_deg: int, n_rings: int) -> np.ndarray:
    """Random spanning tree with a degree cap plus a few ring-closing bonds."""
    deg = [0] * n  # plain lists: the graphs are tiny and this runs once per molecule and per candidate
    bonds = []
    have = set()
    randint, rand = rs.randint, rs.rand
    for a in range(1, n):
        cand = [i for i in range(a) if deg[i] < tree_deg_cap]
        p = cand[randint(0, len(cand))]
        # random orientation: RDKit bonds have no preferred direction either
        bonds.append((p, a) if rand() < 0.5 else (a, p))
        have.add((p, a))  # p < a
        deg[p] += 1
        deg[a] += 1
    for _ in range(n_rings):
        a, b = (int(v) for v in randint(0, n, size=2))
        if a == b or (min(a, b), max(a, b)) in have:
            continue
        if deg[a] >= max_deg or deg[b] >= max_deg:
            continue
        bonds.append((a, b))
        have.add((min(a, b), max(a, b)))
        deg[a] += 1
        deg[b] += 1
    if not bonds:
        return np.zeros((0, 2), dtype=np.int64)
    return np.asarray(bonds, dtype=np.int64)


def make_mol(rs: np.random.RandomState, mean_atoms: float = 40.0, sd_atoms: float = 6.0,
             lo: int = 20, hi: int = 60) -> MolGraph:
    n = int(np.clip(int(round(rs.normal(mean_atoms, sd_atoms))), lo, hi))
    bonds = _random_bonds(rs, n, tree_deg_cap=4, max_deg=MPN_MAX_NB, n_rings=n // 12)
    atom_feats = _one_hot_blocks(rs, n, (MPN_ELEMS, 6, 5, 4), 1)
    nb = bonds.shape[0]
    bond_feats = np.zeros((nb, MPN_BOND_FDIM), dtype=np.float32)
    if nb:
        rows = np.arange(nb)
        bond_feats[rows, rs.randint(0, 5, size=nb)] = 1.0
        bond_feats[:, 5] = (rs.rand(nb) < 0.4).astype(np.float32)
        bond_feats[rows, 6 + rs.randint(0, 6, size=nb)] = 1.0
    return MolGraph(atom_feats, bonds, bond_feats)


def make_tree(rs: np.random.RandomState, mol: MolGraph, vocab_size: int, node_frac: float = 0.55,
              max_nodes: int = 40, deg_cap: int = 6, shape: str = "random") -> Tree:
    """shape "random": every new node hangs on a random earlier one (shallow, bushy trees); "chain": on the node
    before it with probability 0.85 (long paths: many dependency levels, few messages per level)."""
    n_nodes = int(np.clip(int(round(node_frac * mol.n_atoms)), 2, max_nodes))
    nodes = [TreeNode(i + 1, int(rs.randint(0, vocab_size))) for i in range(n_nodes)]
    deg = np.zeros(n_nodes, dtype=np.int64)
    for i in range(1, n_nodes):
        cand = np.flatnonzero(deg[:i] < deg_cap)
        if shape == "chain" and deg[i - 1] < deg_cap and rs.rand() < 0.85:
            p = i - 1
        else:
            p = int(cand[rs.randint(0, len(cand))])
        nodes[p].neighbors.append(nodes[i])
        nodes[i].neighbors.append(nodes[p])
        deg[p] += 1
        deg[i] += 1
    for nd in nodes:
        nd.is_leaf = len(nd.neighbors) == 1
    return Tree(nodes=nodes, mol=mol, smiles="mol")


def _hub_candidate(rs: np.random.RandomState, tree: Tree, node: TreeNode) -> CandGraph:
    """A star whose centre collects the full neighbour table of the candidate encoder (fast_jtnn/jtmpn.py:39:
    15 slots): seven bonds to atoms of neighbouring tree nodes - each brings its bond and the tree message from
    that node - and one bond to an unlabeled atom."""
    n = 9
    bonds = np.asarray([(0, a) if a % 2 else (a, 0) for a in range(1, n)], dtype=np.int64)
    atom_feats = _one_hot_blocks(rs, n, (JTMPN_ELEMS, 6, 5), 1)
    bond_feats = np.zeros((n - 1, JTMPN_BOND_FDIM), dtype=np.float32)
    bond_feats[np.arange(n - 1), rs.randint(0, 5, size=n - 1)] = 1.0
    nid = np.zeros(n, dtype=np.int64)
    nid[0] = node.nid
    nbr_ids = [y.nid for y in node.neighbors]
    for a in range(1, 8):
        nid[a] = nbr_ids[(a - 1) % len(nbr_ids)]
    return CandGraph(atom_feats, bonds, bond_feats, nid, tree.nodes, node)


def make_candidates(rs: np.random.RandomState, tree: Tree, lo_c: int = 2, hi_c: int = 8,
                    lo_atoms: int = 8, hi_atoms: int = 16, hub: bool = False) -> None:
    """Attach 2-8 candidate sub-graphs to every non-leaf node (shape of
    fast_jtnn/datautils.py:150-158: leaves and single-candidate nodes are skipped).  ``hub``: the first
    non-leaf node of the tree gets one more candidate, a star that fills all 15 neighbour slots of its centre."""
    hub_done = not hub
    for node in tree.nodes:
        if node.is_leaf:
            node.cands = []
            node.label = None
            continue
        ncand = int(rs.randint(lo_c, hi_c + 1))
        cands = []
        for _ in range(ncand):
            n = int(rs.randint(lo_atoms, hi_atoms + 1))
            bonds = _random_bonds(rs, n, tree_deg_cap=4, max_deg=5, n_rings=1)
            atom_feats = _one_hot_blocks(rs, n, (JTMPN_ELEMS, 6, 5), 1)
            nb = bonds.shape[0]
            bond_feats = np.zeros((nb, JTMPN_BOND_FDIM), dtype=np.float32)
            rows = np.arange(nb)
            bond_feats[rows, rs.randint(0, 5, size=nb)] = 1.0
            bond_feats[:, 5] = (rs.rand(nb) < 0.4).astype(np.float32)
            # the first third of the atoms belong to the centre node, the rest to
            # its tree neighbours or to nobody (atom-map number 0)
            nid = np.zeros(n, dtype=np.int64)
            n_ctr = max(2, n // 3)
            nid[:n_ctr] = node.nid
            nbr_ids = np.asarray([y.nid for y in node.neighbors], dtype=np.int64)
            pick = rs.randint(0, len(nbr_ids) + 1, size=n - n_ctr)
            nid[n_ctr:] = np.where(pick == len(nbr_ids), 0, nbr_ids[np.minimum(pick, len(nbr_ids) - 1)])
            cands.append(CandGraph(atom_feats, bonds, bond_feats, nid, tree.nodes, node))
        label = cands[int(rs.randint(0, ncand))]
        if not hub_done:
            cands.append(_hub_candidate(rs, tree, node))
            hub_done = True
        node.cands = cands
        node.label = label


def make_batch(seed: int, batch_size: int, vocab_size: int = 385, mean_atoms: float = 40.0,
               with_cands: bool = False, tree_shape: str = "random", cand_hub: bool = False) -> List[Tree]:
    """``batch_size`` synthetic molecules with their junction trees.

    Defaults give the benchmark molecule: about 40 atoms / 42 bonds and about 22
    tree nodes / 42 directed tree messages, neighbour count at most 6.
    ``tree_shape``: "random" (bushy, ~14 dependency levels at the benchmark size), "chain" (path-like, deep) or
    "mixed" (alternating); ``cand_hub``: see ``make_candidates``.
    """
    if tree_shape not in ("random", "chain", "mixed"):
        raise ValueError("tree_shape must be random, chain or mixed")
    rs = np.random.RandomState(seed)
    lo = max(2, int(round(mean_atoms / 2)))
    hi = int(round(mean_atoms * 1.5))
    batch = []
    for i in range(batch_size):
        mol = make_mol(rs, mean_atoms=mean_atoms, sd_atoms=mean_atoms * 0.15, lo=lo, hi=hi)
        shape = tree_shape if tree_shape != "mixed" else ("chain" if i % 2 else "random")
        tree = make_tree(rs, mol, vocab_size, shape=shape)
        if with_cands:
            make_candidates(rs, tree, hub=cand_hub)
        batch.append(tree)
    set_batch_nodeID(batch)
    return batch


def collect_candidates(tree_batch: Sequence[Tree]):
    """Candidate list and owner-molecule index exactly as the training tensorizer
    assembles them (fast_jtnn/datautils.py:150-161)."""
    cands, batch_idx = [], []
    for i, tree in enumerate(tree_batch):
        for node in tree.nodes:
            if node.is_leaf or len(node.cands) == 1:
                continue
            cands.extend(node.cands)
            batch_idx.extend([i] * len(node.cands))
    return cands, np.asarray(batch_idx, dtype=np.int64)


def fill_params(named_params, seed: int, bias_scale: float = 0.0) -> str:
    """Deterministic weights: matrices get Xavier-normal values, vectors are zero
    (the trainer's rule, fast_molvae/vae_train.py:54-58) or, with ``bias_scale``>0,
    small normal values so that bias paths are exercised too.

    Each tensor's stream is keyed by its *name*, so the fill does not depend on
    registration order.  Returns a sha256 over all values for pinning.
    """
    import hashlib

    import torch

    h = hashlib.sha256()
    for name, p in sorted(named_params, key=lambda kv: kv[0]):
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
        shape = tuple(p.shape)
        if len(shape) >= 2:
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            std = float(np.sqrt(2.0 / (fan_in + fan_out)))
            v = (rs.standard_normal(shape) * std).astype(np.float32)
        else:
            v = (rs.standard_normal(shape) * bias_scale).astype(np.float32)
        h.update(name.encode())
        h.update(v.tobytes())
        with torch.no_grad():
            p.copy_(torch.from_numpy(v))
    return h.hexdigest()
