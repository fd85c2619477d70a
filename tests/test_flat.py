"""CPU: the native tensorizers (flat arrays -> reference tables / CSR form) against the numpy statements,
the oracle restatement of the reference's loops and the table route of graphprep."""
import numpy as np
import pytest
import torch

from fast_jtnn_hip import _capi as K
from fast_jtnn_hip import flat, flatprep, graphprep, synthetic
from fast_jtnn_hip import tensorize as T
from oracle import ref_tensorize as R


def _batch(seed, B, atoms=30.0, V=97):
    return synthetic.make_batch(seed, B, vocab_size=V, mean_atoms=atoms, with_cands=True)


@pytest.mark.parametrize("seed,B,atoms", [(21, 40, 40.0), (22, 9, 8.0), (23, 1, 20.0)])
def test_native_tables_equal_numpy_and_oracle(seed, B, atoms):
    trees = _batch(seed, B, atoms)
    mols = [t.mol for t in trees]
    a, b, c = T.mpn_tensorize(mols), T.mpn_tensorize_np(mols), R.mpn_tensorize(mols)
    for x, y, z in zip(a[:4], b[:4], c[:4]):
        assert x.dtype == y.dtype == z.dtype and torch.equal(x, y) and torch.equal(x, z)
    assert a[4] == b[4] == c[4]
    (ha, da), (hc, dc) = T.jtenc_tensorize(trees), R.jtenc_tensorize(trees)
    nodes = [x for t in trees for x in t.nodes]
    hb, db = T.jtenc_tensorize_nodes_np(nodes, ha[4])
    for x, y, z in zip(ha[:4], hb[:4], hc[:4]):
        assert x.dtype == torch.int64 and torch.equal(x, y) and torch.equal(x, z)
    assert ha[4] == hc[4] and da == dc and dict(da) == db and len(da) == len(dc)
    cands, bi = synthetic.collect_candidates(trees)
    ca, cb, cc = T.jtmpn_tensorize(cands, da), T.jtmpn_tensorize_np(cands, dict(da)), R.jtmpn_tensorize(cands, dc)
    for x, y, z in zip(ca[:4], cb[:4], cc[:4]):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert ca[4] == cb[4] == cc[4]
    # the candidate builder also takes a plain dictionary
    cd = T.jtmpn_tensorize(cands, dc)
    assert all(torch.equal(x, y) for x, y in zip(ca[:4], cd[:4]))


def test_selection_threads_and_shard_reuse():
    """A batch is a list of positions in a shard: any order, any thread count, same tables as tensorizing the
    selected objects."""
    trees = _batch(31, 50, 24.0)
    shard = flat.shard_from_trees(trees, with_cands=True)
    rs = np.random.RandomState(0)
    ids = rs.permutation(50)[:17]
    sub = [trees[i] for i in ids]
    # objects of the selection, renumbered the way datautils.tensorize does it per batch
    import copy
    sub = copy.deepcopy(sub)
    synthetic.set_batch_nodeID(sub)
    want_m = R.mpn_tensorize([t.mol for t in sub])
    (want_t, want_d) = R.jtenc_tensorize(sub)
    for th in (1, 3, 8):
        fa, fb, ag, bg, sc = flat.graph_tables(shard.mols, ids, threads=th)
        assert np.array_equal(fa, want_m[0].numpy()) and np.array_equal(fb, want_m[1].numpy())
        assert np.array_equal(ag, want_m[2].numpy()) and np.array_equal(bg, want_m[3].numpy())
        assert [tuple(r) for r in sc.tolist()] == want_m[4]
        (fn, fm, ng, mg, tsc), (src, dst) = flat.tree_tables(shard.trees, ids, threads=th)
        assert np.array_equal(fn, want_t[0].numpy()) and np.array_equal(fm, want_t[1].numpy())
        assert np.array_equal(ng, want_t[2].numpy()) and np.array_equal(mg, want_t[3].numpy())
        assert [tuple(r) for r in tsc.tolist()] == want_t[4]
        assert dict(flat.MessDict(src, dst, fn.shape[0])) == want_d
    # candidates of the selection
    cand_ids, batch_idx, node_base, seg_len, label_pos = flat.select_candidates(shard.trees, ids)
    cands, bi = synthetic.collect_candidates(sub)
    assert np.array_equal(batch_idx, bi)
    want_c = R.jtmpn_tensorize(cands, want_d)
    mess = flat.candidate_mess(src, dst, fn.shape[0], node_base)
    got_c = flat.graph_tables(shard.cands, cand_ids, mess=mess, max_nb=flat.JTMPN_MAX_NB, threads=4)
    for g, w in zip(got_c[:4], want_c[:4]):
        assert np.array_equal(g, w.numpy())
    # loss segments as JTNNVAE.assm derives them from the objects (fast_jtnn/jtnn_vae.py:116-135)
    sl, lp, tot = [], [], 0
    for t in sub:
        for node in t.nodes:
            if len(node.cands) > 1 and not node.is_leaf:
                lp.append(tot + node.cands.index(node.label))
                sl.append(len(node.cands))
                tot += len(node.cands)
    assert seg_len.tolist() == sl and label_pos.tolist() == lp and tot == len(cands)


def test_csr_form_is_the_tables_without_their_padding():
    trees = _batch(41, 33, 30.0)
    shard = flat.shard_from_trees(trees, with_cands=True)
    fa, fb, ag, bg, sc = flat.graph_tables(shard.mols)
    for train in (False, True):
        o = flat.graph_csr(shard.mols, train=train, threads=3)
        assert np.array_equal(o["fatoms"], fa.astype(np.uint8)) and np.array_equal(o["fbonds"], fb.astype(np.uint8))
        for tab, rp, col, nnz in ((ag, "a_rowptr", "a_col", "a_nnz"), (bg, "b_rowptr", "b_col", "b_nnz")):
            w_rp, w_col = graphprep.padded_to_csr_np(tab)
            assert np.array_equal(o[rp], w_rp) and np.array_equal(o[col][:o[nnz]], w_col) and o[nnz] == w_col.size
            if train:
                t_rp, t_col = graphprep.transpose_csr_np(w_rp, w_col, bg.shape[0])
                assert np.array_equal(o[rp.replace("_", "t_", 1)], t_rp)
                assert np.array_equal(o[col.replace("_", "t_", 1)][:o[nnz]], t_col)
        assert np.array_equal(o["seg"], sc) and o["a_tree_nnz"] == 0 == o["b_tree_nnz"]
        assert np.array_equal(o["seg_of"], np.repeat(np.arange(len(trees)), sc[:, 1]))
    (fn, fm, ng, mg, tsc), (src, dst) = flat.tree_tables(shard.trees)
    t = flat.tree_csr(shard.trees, threads=2)
    m_rp, m_col = graphprep.padded_to_csr_np(mg)
    n_rp, n_col = graphprep.padded_to_csr_np(ng)
    assert np.array_equal(t["m_rowptr"], m_rp) and np.array_equal(t["m_col"][:t["m_nnz"]], m_col)
    assert np.array_equal(t["n_rowptr"], n_rp) and np.array_equal(t["n_col"][:n_col.size], n_col)
    assert np.array_equal(t["wid"][:fn.size], fn) and np.array_equal(t["src"], fm) and np.array_equal(t["src"], src)
    assert np.array_equal(t["roots"], tsc[:, 0])
    sf = K.prep_stable_from(m_rp, m_col, 2 ** 30)
    assert np.array_equal(t["sf"], sf) and t["max_sf"] == int(sf.max())
    # candidate graphs: two-source index space, tree entries counted
    cand_ids, batch_idx, node_base, _, _ = flat.select_candidates(shard.trees)
    mess = flat.candidate_mess(src, dst, fn.shape[0], node_base)
    cfa, cfb, cag, cbg, csc = flat.graph_tables(shard.cands, cand_ids, mess=mess, max_nb=flat.JTMPN_MAX_NB)
    o = flat.graph_csr(shard.cands, cand_ids, mess=mess, max_nb=flat.JTMPN_MAX_NB, train=True, threads=5)
    tm = fm.shape[0]
    for tab, rp, col, nnz, tn in ((cag, "a_rowptr", "a_col", "a_nnz", "a_tree_nnz"),
                                  (cbg, "b_rowptr", "b_col", "b_nnz", "b_tree_nnz")):
        w_rp, w_col = graphprep.padded_to_csr_np(tab)
        assert np.array_equal(o[rp], w_rp) and np.array_equal(o[col][:o[nnz]], w_col)
        assert o[tn] == int((w_col < tm).sum()) and o[tn] > 0
        t_rp, t_col = graphprep.transpose_csr_np(w_rp, w_col, tm + cbg.shape[0])
        assert np.array_equal(o[rp.replace("_", "t_", 1)], t_rp)
        assert np.array_equal(o[col.replace("_", "t_", 1)][:o[nnz]], t_col)


def test_edge_cases_and_errors():
    # a molecule without bonds, a single-node tree
    mol = synthetic.MolGraph(np.eye(1, 39, dtype=np.float32), np.zeros((0, 2), np.int64), np.zeros((0, 12), np.float32))
    fa, fb, ag, bg, sc = T.mpn_tensorize([mol])
    assert fa.shape == (1, 39) and fb.shape == (1, 51) and ag.shape == (1, 6) and bg.shape == (1, 6) and sc == [(0, 1)]
    node = synthetic.TreeNode(1, 3)
    node.idx = 0
    (fn, fm, ng, mg, s2), md = T.jtenc_tensorize_nodes([node], [(0, 1)])
    ref, rd = R.jtenc_tensorize_nodes([node], [(0, 1)])
    assert all(torch.equal(x, y) for x, y in zip((fn, fm, ng, mg), ref[:4])) and len(md) == 0 == len(rd)
    # the seventh neighbour of an atom: the reference's agraph[a, i] = b raises IndexError
    bonds = np.asarray([[0, i] for i in range(1, 8)], dtype=np.int64)
    star = synthetic.MolGraph(np.zeros((8, 39), np.float32), bonds, np.zeros((7, 12), np.float32))
    with pytest.raises(IndexError):
        T.mpn_tensorize([star])
    with pytest.raises(IndexError):
        R.mpn_tensorize([star])
    # an atom index outside the molecule
    bad = synthetic.MolGraph(np.zeros((2, 39), np.float32), np.asarray([[0, 5]], dtype=np.int64), np.zeros((1, 12), np.float32))
    with pytest.raises(IndexError):
        T.mpn_tensorize([bad])
    # feature values that are not whole numbers take the numpy route and keep their values
    frac = synthetic.MolGraph(np.full((2, 39), 0.5, np.float32), np.asarray([[0, 1]], dtype=np.int64), np.full((1, 12), 0.25, np.float32))
    fa, fb, _, _, _ = T.mpn_tensorize([frac])
    assert float(fa[0, 0]) == 0.5 and float(fb[1, 50]) == 0.25
    # selection outside the shard
    shard = flat.shard_from_trees(_batch(5, 3, 10.0))
    with pytest.raises(IndexError):
        flat.graph_tables(shard.mols, np.asarray([0, 3]))
    with pytest.raises(IndexError):
        flat.tree_tables(shard.trees, np.asarray([-1]))


def test_flat_route_prepares_what_the_table_route_prepares():
    """flatprep (no tables) and graphprep (from the reference's tables) end in the same device-side arrays."""
    trees = _batch(51, 20, 26.0)
    shard = flat.shard_from_trees(trees, with_cands=True)
    dev = torch.device("cpu")
    jt, md = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    V, depth = 97, 20

    def same_csr(a, b):
        assert a.n_rows == b.n_rows and a.nnz == b.nnz
        assert torch.equal(a.rowptr.cpu(), b.rowptr.cpu()) and torch.equal(a.col.cpu()[:a.nnz], b.col.cpu()[:b.nnz])

    want = graphprep.prepare_mpn(mp, dev, 0, True)
    got = flatprep.graph_prep(shard.mols, None, dev, train=True, threads=2)
    for f in ("n_atoms", "n_bonds", "n_tree", "n_seg", "atom_fdim", "bond_in_fdim", "a_tree_nnz", "b_tree_nnz"):
        assert getattr(want, f) == getattr(got, f), f
    assert torch.equal(want.float_rows("fatoms"), got.float_rows("fatoms"))
    assert torch.equal(want.float_rows("fbonds"), got.float_rows("fbonds"))
    for f in ("acsr", "bcsr", "acsr_t", "bcsr_t"):
        same_csr(getattr(want, f), getattr(got, f))
    assert torch.equal(want.seg.cpu(), got.seg.cpu()) and torch.equal(want.seg_of.cpu(), got.seg_of.cpu())

    wt = graphprep.prepare_tree(jt, dev, V, True)
    gt, o = flatprep.tree_prep(shard.trees, None, dev, V, depth, train=True, all_messages=True)
    for f in ("n_nodes", "n_mess", "n_trees", "n_words"):
        assert getattr(wt, f) == getattr(gt, f), f
    for f in ("mcsr", "mcsr_t", "rcsr", "rcsr_t", "word_csr", "word_csr_t", "ident", "key_m_pieces", "key_m",
              "key_r_pieces", "key_r"):
        same_csr(getattr(wt, f), getattr(gt, f))
    for f in ("words", "wid_m", "wid_root"):
        assert torch.equal(getattr(wt, f), getattr(gt, f))
    assert np.array_equal(wt.host["sf_true"], gt.host["sf_true"])
    # candidates
    cands, bi = synthetic.collect_candidates(trees)
    cj = T.jtmpn_tensorize(cands, md)
    wc = graphprep.prepare_mpn(cj, dev, int(jt[1].shape[0]), True)
    cand_ids, batch_idx, node_base, _, _ = flat.select_candidates(shard.trees)
    mess = flat.candidate_mess(o["src"], o["dst"], o["n_nodes"], node_base)
    gc = flatprep.graph_prep(shard.cands, cand_ids, dev, mess, flat.JTMPN_MAX_NB, True)
    for f in ("n_atoms", "n_bonds", "n_tree", "n_seg", "a_tree_nnz", "b_tree_nnz"):
        assert getattr(wc, f) == getattr(gc, f), f
    for f in ("acsr", "bcsr", "acsr_t", "bcsr_t"):
        same_csr(getattr(wc, f), getattr(gc, f))
    assert torch.equal(wc.float_rows("fbonds"), gc.float_rows("fbonds"))


def test_deep_tree_takes_the_table_route():
    """A chain longer than the iteration count cannot be computed once per message: flatprep hands back the
    reference's tables for it (the encoder then splits such trees off)."""
    nodes = [synthetic.TreeNode(i + 1, i % 7) for i in range(30)]
    for a, b in zip(nodes[:-1], nodes[1:]):
        a.neighbors.append(b)
        b.neighbors.append(a)
    mol = synthetic.make_mol(np.random.RandomState(1))
    tree = synthetic.Tree(nodes=nodes, mol=mol)
    synthetic.set_batch_nodeID([tree])
    shard = flat.shard_from_trees([tree])
    prep, o = flatprep.tree_prep(shard.trees, None, torch.device("cpu"), 7, 20, train=False)
    assert prep is None and o["max_sf"] == 29
    holder, _ = flatprep._tree_holder(shard.trees, None, torch.device("cpu"), 7, 20, False, False, None)
    want, _ = R.jtenc_tensorize([tree])
    assert all(torch.equal(x, y) for x, y in zip(holder[:4], want[:4])) and holder[4] == want[4]


def test_shard_file_round_trip(tmp_path):
    trees = _batch(61, 7, 14.0)
    shard = flat.shard_from_trees(trees, with_cands=True)
    props = np.arange(14, dtype=np.float32).reshape(7, 2)
    path = str(tmp_path / "set.npz")
    flat.save_shard(path, shard, props)
    back, p2 = flat.load_shard(path)
    assert np.array_equal(p2, props) and back.n == 7
    for a, b in ((shard.trees, back.trees), (shard.mols, back.mols), (shard.cands, back.cands)):
        for k, v in vars(a).items():
            assert np.array_equal(v, getattr(b, k)), k
    assert all(np.array_equal(x, y) for x, y in zip(flat.graph_tables(shard.mols), flat.graph_tables(back.mols)))
