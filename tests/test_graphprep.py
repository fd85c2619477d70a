"""CPU: padded neighbour tables -> CSR, transposes and key lists (host index bookkeeping)."""
import numpy as np

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
