"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's *index logic*: how atoms, directed bonds, tree
nodes and tree messages are numbered and how the neighbour tables are filled.
Plain Python loops in the reference's order, on abstract graphs (the reference
takes SMILES through RDKit; everything after the RDKit calls is restated here).

Follows:
  fast_jtnn/mpn.py:110-161      MPN.tensorize
  fast_jtnn/jtnn_enc.py:44-92   JTNNEncoder.tensorize / tensorize_nodes
  fast_jtnn/jtmpn.py:113-183    JTMPN.tensorize
  fast_jtnn/datautils.py:150-161 candidate list / batch_idx

Pinned by: tests/golden.  All tables stored there were built by the reference's own
tensorizers (JTNNEncoder.tensorize as it is; MPN.tensorize and JTMPN.tensorize with
their RDKit accessors answered by duck-typed molecule objects, see
tests/golden/make_golden.py), and the generator asserts that this restatement
produces the same tables bit for bit.
"""
import torch

MPN_MAX_NB = 6  # fast_jtnn/mpn.py:37
JTMPN_MAX_NB = 15  # fast_jtnn/jtmpn.py:39


def mpn_tensorize(mols):
    """fast_jtnn/mpn.py:110-161 with ``mol`` = object with atom_feats/bonds/bond_feats."""
    fdim = mols[0].atom_feats.shape[1] + mols[0].bond_feats.shape[1]
    padding = torch.zeros(fdim)
    fatoms, fbonds = [], [padding]  # bond ids are 1-based (mpn.py:113)
    in_bonds, all_bonds = [], [(-1, -1)]
    scope = []
    total_atoms = 0
    for mol in mols:
        n_atoms = mol.atom_feats.shape[0]
        for a in range(n_atoms):
            fatoms.append(torch.from_numpy(mol.atom_feats[a]))
            in_bonds.append([])
        for bi in range(mol.bonds.shape[0]):
            x = int(mol.bonds[bi, 0]) + total_atoms
            y = int(mol.bonds[bi, 1]) + total_atoms
            bf = torch.from_numpy(mol.bond_feats[bi])
            b = len(all_bonds)  # mpn.py:132-135
            all_bonds.append((x, y))
            fbonds.append(torch.cat([fatoms[x], bf], 0))
            in_bonds[y].append(b)
            b = len(all_bonds)  # mpn.py:137-140
            all_bonds.append((y, x))
            fbonds.append(torch.cat([fatoms[y], bf], 0))
            in_bonds[x].append(b)
        scope.append((total_atoms, n_atoms))
        total_atoms += n_atoms

    total_bonds = len(all_bonds)
    fatoms = torch.stack(fatoms, 0)
    fbonds = torch.stack(fbonds, 0)
    agraph = torch.zeros(total_atoms, MPN_MAX_NB).long()
    bgraph = torch.zeros(total_bonds, MPN_MAX_NB).long()
    for a in range(total_atoms):  # mpn.py:151-153
        for i, b in enumerate(in_bonds[a]):
            agraph[a, i] = b
    for b1 in range(1, total_bonds):  # mpn.py:155-159
        x, y = all_bonds[b1]
        for i, b2 in enumerate(in_bonds[x]):
            if all_bonds[b2][0] != y:
                bgraph[b1, i] = b2
    return fatoms, fbonds, agraph, bgraph, scope


def jtenc_tensorize(tree_batch):
    """fast_jtnn/jtnn_enc.py:44-52."""
    node_batch, scope = [], []
    for tree in tree_batch:
        scope.append((len(node_batch), len(tree.nodes)))
        node_batch.extend(tree.nodes)
    return jtenc_tensorize_nodes(node_batch, scope)


def jtenc_tensorize_nodes(node_batch, scope):
    """fast_jtnn/jtnn_enc.py:54-92."""
    messages, mess_dict = [None], {}
    fnode = []
    for x in node_batch:
        fnode.append(x.wid)
        for y in x.neighbors:
            mess_dict[(x.idx, y.idx)] = len(messages)
            messages.append((x, y))

    node_graph = [[] for _ in range(len(node_batch))]
    mess_graph = [[] for _ in range(len(messages))]
    fmess = [0] * len(messages)
    for x, y in messages[1:]:
        mid1 = mess_dict[(x.idx, y.idx)]
        fmess[mid1] = x.idx
        node_graph[y.idx].append(mid1)
        for z in y.neighbors:
            if z.idx == x.idx:
                continue
            mid2 = mess_dict[(y.idx, z.idx)]
            mess_graph[mid2].append(mid1)

    max_len = max([len(t) for t in node_graph] + [1])
    for t in node_graph:
        t.extend([0] * (max_len - len(t)))
    max_len = max([len(t) for t in mess_graph] + [1])
    for t in mess_graph:
        t.extend([0] * (max_len - len(t)))

    return (
        torch.LongTensor(fnode),
        torch.LongTensor(fmess),
        torch.LongTensor(node_graph),
        torch.LongTensor(mess_graph),
        scope,
    ), mess_dict


def collect_candidates(tree_batch):
    """fast_jtnn/datautils.py:150-161."""
    cands, batch_idx = [], []
    for i, tree in enumerate(tree_batch):
        for node in tree.nodes:
            if node.is_leaf or len(node.cands) == 1:
                continue
            cands.extend(node.cands)
            batch_idx.extend([i] * len(node.cands))
    return cands, torch.LongTensor(batch_idx)


def jtmpn_tensorize(cand_batch, mess_dict):
    """fast_jtnn/jtmpn.py:113-183 with ``cand`` = object with
    atom_feats/bonds/bond_feats/atom_nid/all_nodes."""
    fatoms, fbonds = [], []
    in_bonds, all_bonds = [], []
    total_atoms = 0
    total_mess = len(mess_dict) + 1  # includes the zero row (jtmpn.py:118)
    scope = []
    for cand in cand_batch:
        all_nodes = cand.all_nodes
        n_atoms = cand.atom_feats.shape[0]
        for a in range(n_atoms):
            fatoms.append(torch.from_numpy(cand.atom_feats[a]))
            in_bonds.append([])
        for bi in range(cand.bonds.shape[0]):
            a1, a2 = int(cand.bonds[bi, 0]), int(cand.bonds[bi, 1])
            x = a1 + total_atoms
            y = a2 + total_atoms
            x_nid, y_nid = int(cand.atom_nid[a1]), int(cand.atom_nid[a2])
            x_bid = all_nodes[x_nid - 1].idx if x_nid > 0 else -1  # jtmpn.py:141-142
            y_bid = all_nodes[y_nid - 1].idx if y_nid > 0 else -1
            bf = torch.from_numpy(cand.bond_feats[bi])

            b = total_mess + len(all_bonds)  # jtmpn.py:146-149
            all_bonds.append((x, y))
            fbonds.append(torch.cat([fatoms[x], bf], 0))
            in_bonds[y].append(b)

            b = total_mess + len(all_bonds)  # jtmpn.py:151-154
            all_bonds.append((y, x))
            fbonds.append(torch.cat([fatoms[y], bf], 0))
            in_bonds[x].append(b)

            if x_bid >= 0 and y_bid >= 0 and x_bid != y_bid:  # jtmpn.py:156-162
                if (x_bid, y_bid) in mess_dict:
                    in_bonds[y].append(mess_dict[(x_bid, y_bid)])
                if (y_bid, x_bid) in mess_dict:
                    in_bonds[x].append(mess_dict[(y_bid, x_bid)])
        scope.append((total_atoms, n_atoms))
        total_atoms += n_atoms

    total_bonds = len(all_bonds)
    fatoms = torch.stack(fatoms, 0)
    fbonds = torch.stack(fbonds, 0)
    agraph = torch.zeros(total_atoms, JTMPN_MAX_NB).long()
    bgraph = torch.zeros(total_bonds, JTMPN_MAX_NB).long()
    for a in range(total_atoms):  # jtmpn.py:173-175
        for i, b in enumerate(in_bonds[a]):
            agraph[a, i] = b
    for b1 in range(total_bonds):  # jtmpn.py:177-181
        x, y = all_bonds[b1]
        for i, b2 in enumerate(in_bonds[x]):
            if b2 < total_mess or all_bonds[b2 - total_mess][0] != y:
                bgraph[b1, i] = b2
    return fatoms, fbonds, agraph, bgraph, scope
