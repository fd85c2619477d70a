"""CPU: padded neighbour tables -> CSR, transposes and key lists (host index bookkeeping)."""
import numpy as np
import pytest

from fast_jtnn_hip import graphprep as G


def brute_transpose(table, n_src):
    lists = [[] for _ in range(n_src)]
    for i in range(table.shape[0]):
        for k in range(table.shape[1]):
            j = int(table[i, k])
            if j:
                lists[j].append(i)
    return lists


def test_padded_to_csr_keeps_slot_order_and_drops_interleaved_zeros():
    t = np.array([[0, 0, 0], [5, 0, 2], [0, 7, 7], [1, 2, 3]], dtype=np.int64)
    rp, col = G.padded_to_csr(t)
    assert rp.tolist() == [0, 0, 2, 4, 7]
    assert col.tolist() == [5, 2, 7, 7, 1, 2, 3]


def test_transpose_matches_brute_force():
    rs = np.random.RandomState(0)
    for n, k, n_src in ((50, 6, 50), (200, 15, 350), (1, 1, 3)):
        t = rs.randint(0, n_src, size=(n, k)) * (rs.rand(n, k) < 0.4)
        rp, col = G.padded_to_csr(t)
        trp, tcol = G.transpose_csr(rp, col, n_src)
        want = brute_transpose(t, n_src)
        assert trp[0] == 0 and trp[-1] == col.shape[0]
        for j in range(n_src):
            assert tcol[trp[j]:trp[j + 1]].tolist() == want[j]
        assert trp[1] - trp[0] == 0  # nobody reads the padding row


def test_chunk_lists():
    rp = np.array([0, 0, 5, 5, 70, 71], dtype=np.int64)
    prp, krp = G.chunk_lists(rp, 32)
    # keys: empty, 5, empty, 65 (3 pieces), 1
    assert krp.tolist() == [0, 1, 2, 3, 6, 7]
    assert prp.tolist() == [0, 0, 5, 5, 37, 69, 70, 71]


def test_scope_arrays():
    seg, seg_of = G.scope_arrays([(0, 2), (2, 3)], 5)
    assert seg_of.tolist() == [0, 0, 1, 1, 1]
    seg, seg_of = G.scope_arrays([(3, 2), (0, 2)], 6)  # unordered, with a gap
    assert seg_of.tolist() == [1, 1, -1, 0, 0, -1]
    import pytest
    with pytest.raises(IndexError):
        G.scope_arrays([(0, 7)], 5)


def test_range_check():
    import pytest
    with pytest.raises(IndexError):
        G.check_range(np.array([1, 9]), 9, "x")


def test_native_prep_matches_numpy_statements():
    """The single-pass C routines against their numpy statements, ragged and empty inputs included."""
    from fast_jtnn_hip import _capi as K
    rs = np.random.RandomState(11)
    for rows, kmax, n_src in ((0, 6, 5), (1, 1, 2), (57, 6, 57), (400, 15, 900), (33, 9, 33)):
        t = rs.randint(0, max(n_src, 1), size=(rows, kmax)).astype(np.int64)
        t[rs.rand(rows, kmax) < 0.5] = 0
        rp, col = G.padded_to_csr(t, n_src)
        rp_np, col_np = G.padded_to_csr_np(t)
        assert rp.dtype == np.int32 and np.array_equal(rp, rp_np) and np.array_equal(col, col_np)
        trp, tcol = G.transpose_csr(rp, col, n_src)
        trp_np, tcol_np = G.transpose_csr_np(rp_np, col_np, n_src)
        assert np.array_equal(trp, trp_np) and np.array_equal(tcol, tcol_np)
        if rows == n_src:  # a message graph: rows read rows
            for T in (0, 1, 3, 20):
                sf = K.prep_stable_from(rp, col, T)
                assert np.array_equal(sf, G._levels(rp_np, col_np, rows, max(T, 1), 1))
        keys = rs.randint(0, 7, size=rows).astype(np.int64)
        kp, order = K.prep_group_by_key(keys, 7)
        assert np.array_equal(order, np.argsort(keys, kind="stable"))
        assert np.array_equal(np.diff(kp), np.bincount(keys, minlength=7))
    with pytest.raises(IndexError):
        G.padded_to_csr(np.array([[1, 9]], dtype=np.int64), 9)
    with pytest.raises(IndexError):
        G.padded_to_csr(np.array([[-1, 0]], dtype=np.int64), 9)


def test_feature_rows_become_bytes_only_when_that_loses_nothing():
    from fast_jtnn_hip import _capi as K

    rs = np.random.RandomState(0)
    x = (rs.rand(9001, 50) < 0.08).astype(np.float32)
    x[17, 3], x[9000, 49] = 255.0, 6.0
    out = np.empty(x.shape, dtype=np.uint8)
    assert K.prep_features_to_u8(x, out) and np.array_equal(out.astype(np.float32), x)
    for bad in (0.5, -1.0, 256.0, np.nan, np.inf, -0.0, 1e-45):
        for pos in ((0, 0), (4500, 25), (9000, 49)):  # first block, a middle one, the very last value
            y = x.copy()
            y[pos] = bad
            assert not K.prep_features_to_u8(y, out), (bad, pos)
    assert K.prep_features_to_u8(np.zeros((0, 39), dtype=np.float32), np.empty((0, 39), dtype=np.uint8))
    # large enough that the pass is cut into ranges converted side by side: each range reports its own misfit
    big = (rs.rand(130001, 50) < 0.08).astype(np.float32)
    out = np.zeros(big.shape, dtype=np.uint8)
    assert K.prep_features_to_u8(big, out) and np.array_equal(out.astype(np.float32), big)
    for pos in ((0, 1), (40000, 2), (70000, 3), (100000, 4), (130000, 49)):
        y = big.copy()
        y[pos] = 0.25
        assert not K.prep_features_to_u8(y, out), pos
    import ctypes
    assert K.lib.jtvae_prep_features_to_u8(None, 4, None, ctypes.byref(ctypes.c_int32())) == K.E_NULL
    assert K.lib.jtvae_prep_features_to_u8(None, -1, None, ctypes.byref(ctypes.c_int32())) == K.E_SHAPE


def test_compute_once_schedule_covers_every_message_exactly_once():
    """Mode "once": apart from iteration 0 (which evaluates every message on empty inputs), a message is
    updated in exactly one iteration, the one in which it settles, and reads only settled messages."""
    from fast_jtnn_hip import synthetic, tensorize as T
    trees = synthetic.make_batch(3, 40, vocab_size=30)
    jt, _ = T.jtenc_tensorize(trees)
    mg = jt[3].numpy()
    nm = mg.shape[0]
    rp, col = G.padded_to_csr(mg, nm)
    trp, tcol = G.transpose_csr(rp, col, nm)
    Tn = 20
    once = G.build_tree_schedule(rp, col, trp, tcol, nm, Tn, "cpu", mode="once")
    level = G.build_tree_schedule(rp, col, trp, tcol, nm, Tn, "cpu", mode="level")
    assert once.mode == "once"
    sf = once.sf.numpy()
    assert np.array_equal(sf, level.sf.numpy())
    seen = np.zeros(nm, dtype=np.int64)
    seen[sf == 1] += 1  # settled by iteration 0
    for t in range(1, Tn):
        rows = once.F[t].rows.numpy() if once.F[t].rows is not None else np.zeros(0, np.int64)
        assert once.F[t].n == rows.size and np.all(sf[rows] == t + 1)
        seen[rows] += 1
        for m in rows[:50]:
            assert np.all(sf[col[rp[m]:rp[m + 1]]] <= t)  # inputs have settled
        urows = once.U[t].rows.numpy() if once.U[t].rows is not None else np.zeros(0, np.int64)
        assert np.all(sf[urows] == t)
        assert once.F[t].n <= level.F[t].n
    assert np.all(seen == 1)
    # a batch deeper than the iteration count falls back to the iteration-by-iteration form
    shallow_T = int(sf.max()) - 1
    assert G.build_tree_schedule(rp, col, trp, tcol, nm, shallow_T, "cpu", mode="once").mode == "level"


def test_native_backward_plan_equals_its_numpy_statement():
    """Backward plan of the compute-once schedule: the native single pass against the numpy statement, with all
    messages seeded and with the root-feeding messages only; junction trees and an edge-free graph."""
    from fast_jtnn_hip import _capi as K, synthetic, tensorize as T
    cases = []
    for seed, n_trees in ((3, 40), (4, 1), (5, 7)):
        jt, _ = T.jtenc_tensorize(synthetic.make_batch(seed, n_trees, vocab_size=30))
        mg, ng = jt[3].numpy(), jt[2].numpy()
        roots = np.asarray([st for st, _ in jt[4]], dtype=np.int64)
        cases.append((mg, G.padded_to_csr(ng[roots], mg.shape[0])[1]))
    cases.append((np.zeros((5, 2), dtype=np.int64), np.array([2, 4], dtype=np.int32)))
    for mg, root_mess in cases:
        nm = mg.shape[0]
        rp, col = G.padded_to_csr(mg, nm)
        trp, tcol = G.transpose_csr(rp, col, nm)
        deg = np.diff(rp).astype(np.int64)
        for seeds in (None, root_mess, root_mess[:1], root_mess[:0]):
            got = K.prep_backward_once(rp, col, seeds)
            want = G.backward_once_plan_np(rp.astype(np.int64), col.astype(np.int64), trp.astype(np.int64),
                                           tcol.astype(np.int64), deg, nm, seeds)
            for k in ("act", "order", "lev_off", "n_dep", "deg_sum", "tdeg_sum"):
                assert np.array_equal(np.asarray(got[k]).astype(np.int64), np.asarray(want[k]).astype(np.int64)), k
    with pytest.raises(IndexError):  # a cycle has no compute-once order
        K.prep_backward_once(np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), None)


def test_native_decoder_plan_equals_the_walk_over_tree_objects():
    """Trace tables of the teacher-forced decoder: the native walk over the junction-tree holder against the
    plain-Python walk over the tree objects (encoder numbering of the messages), field by field."""
    import numpy as np

    from fast_jtnn_hip import synthetic, tensorize as T
    from fast_jtnn_hip.jtnn_dec import decoder_plan, decoder_plan_from_holder

    for seed, B, atoms in ((1, 1, 30.0), (2, 17, 12.0), (3, 40, 40.0)):
        trees = synthetic.make_batch(seed, B, vocab_size=50, mean_atoms=atoms)
        if seed == 2:  # a single-node tree in the middle of the batch
            trees[5] = synthetic.Tree([synthetic.TreeNode(1, 7)], trees[5].mol)
            synthetic.set_batch_nodeID(trees)
        holder, mess_dict = T.jtenc_tensorize(trees)
        a, b = decoder_plan(trees, mess_dict), decoder_plan_from_holder(holder)
        assert (a.n_trees, a.n_mess) == (b.n_trees, b.n_mess)
        for f in ("mess_word", "dep_rp", "dep_col", "stop_rp", "stop_col", "stop_word", "stop_ctx", "stop_target",
                  "pred_mess", "pred_ctx", "pred_target"):
            x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
            if f == "mess_word":  # entry 0 is the padding message: any valid word
                x, y = x[1:], y[1:]
            assert np.array_equal(x, y), (seed, f)


def test_deep_trees_are_split_off_the_batch():
    """One chain longer than the iteration count among ordinary trees: the holder splits into two valid
    holders whose tables are those of tensorizing the two groups on their own."""
    import numpy as np

    from fast_jtnn_hip import graphprep, synthetic, tensorize as T

    trees = synthetic.make_batch(2, 9, vocab_size=30, mean_atoms=14.0)
    chain = [synthetic.TreeNode(i + 1, i % 30) for i in range(12)]
    for a, b in zip(chain[:-1], chain[1:]):
        a.neighbors.append(b); b.neighbors.append(a)
    trees.insert(4, synthetic.Tree(chain, trees[0].mol))
    synthetic.set_batch_nodeID(trees)
    holder, _ = T.jtenc_tensorize(trees)
    assert graphprep.split_deep_trees(holder, 20) is None  # everything settles within 20 iterations
    (sh, sh_trees, sh_mess), (dp, dp_trees, dp_mess) = graphprep.split_deep_trees(holder, 6)
    assert dp_trees.tolist() == [4] and sh_trees.tolist() == [0, 1, 2, 3, 5, 6, 7, 8, 9]
    assert sorted(sh_mess.tolist() + dp_mess.tolist()) == list(range(1, holder[1].shape[0]))
    for sub, ids in ((sh, sh_trees), (dp, dp_trees)):
        part = [trees[i] for i in ids.tolist()]
        synthetic.set_batch_nodeID(part)
        want, _ = T.jtenc_tensorize(part)
        assert np.array_equal(sub[0].numpy(), want[0].numpy()) and np.array_equal(sub[1].numpy(), want[1].numpy())
        k = want[2].shape[1]
        assert np.array_equal(sub[2].numpy()[:, :k], want[2].numpy()) and not sub[2].numpy()[:, k:].any()
        k = want[3].shape[1]
        assert np.array_equal(sub[3].numpy()[:, :k], want[3].numpy()) and not sub[3].numpy()[:, k:].any()
        assert sub[4] == want[4]
    synthetic.set_batch_nodeID(trees)


def test_forward_once_plan_matches_numpy():
    """jtvae_prep_forward_once against the grouping stated in numpy (stable sort by capped settling iteration)."""
    from fast_jtnn_hip import _capi as K
    from fast_jtnn_hip import synthetic
    from fast_jtnn_hip import tensorize as T

    trees = synthetic.make_batch(9, 40, vocab_size=31, mean_atoms=30.0)
    mg = T.jtenc_tensorize(trees)[0][3].numpy()
    rp, col = G.padded_to_csr(mg)
    sf_true = K.prep_stable_from(rp, col, 2 ** 30)
    for depth in (1, 3, 20):
        sf, order, keyptr, u_rows, u_ptr, nnz_ptr = K.prep_forward_once(rp, col, sf_true, depth)
        cap = max(depth, 1)
        want_sf = np.minimum(sf_true, cap)
        assert np.array_equal(sf, want_sf)
        want_order = np.argsort(want_sf, kind="stable")
        assert np.array_equal(order, want_order)
        counts = np.bincount(want_sf, minlength=cap + 1)
        assert np.array_equal(keyptr, np.concatenate([[0], np.cumsum(counts)]))
        read = np.bincount(col, minlength=mg.shape[0])[want_order] > 0
        assert np.array_equal(u_rows, want_order[read])
        assert np.array_equal(u_ptr, np.concatenate([[0], np.cumsum(read)])[keyptr])
        deg = np.diff(rp)[want_order]
        assert np.array_equal(nnz_ptr, np.concatenate([[0], np.cumsum(deg)])[keyptr])
