"""CPU: the C-ABI library loads and exports every symbol include/jtvae_hip.h declares
(no compute calls - there is no GPU here), and rejects bad arguments before any launch."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jtvae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jtvae_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from fast_jtnn_hip import _capi

    names = declared_symbols()
    assert len(names) >= 18
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libjtvae_hip.so does not export %s" % n
    # the ctypes table covers exactly the header
    assert sorted(_capi.SIGNATURES) == names
    assert _capi.lib.jtvae_abi_version() == _capi.ABI_VERSION


def test_argument_checks_happen_before_launch():
    from fast_jtnn_hip import _capi as K

    lib = K.lib
    # NULL pointers, bad shapes, misaligned row buffers are refused with a negative code
    assert lib.jtvae_gemm_nt(None, 4, None, 4, None, 4, 8, 4, 4, None, None) == -1
    assert lib.jtvae_gemm_nt(16, 4, 16, 4, 16, 4, -1, 4, 4, None, None) == -2
    assert lib.jtvae_gemm_nt(16, 2, 16, 4, 16, 4, 8, 4, 4, None, None) == -2  # lda < K
    assert lib.jtvae_csr_gather_sum(16, 16, 1, None, 16, 16, 4, 450, 450, None, 0, None, 0, None) == -3  # ld % 4
    assert lib.jtvae_csr_gather_sum(16, 20, 1, None, 16, 16, 4, 450, 452, None, 0, None, 0, None) == -3  # base
    assert lib.jtvae_gru_iter0(None, None, None, None, None, None, 4, 450, 452, 0, None) == -1
    assert lib.jtvae_gemm_tn(16, 452, None, 16, 452, None, 16, 450, None, 100, 450, 450, 0, 16, 8, None) == -5  # workspace
    assert lib.jtvae_gemm_tn_workspace_bytes(100000, 450, 450) > 0
    assert b"NULL" in lib.jtvae_error_string(-1)
    # zero-sized problems are accepted without touching the device
    assert lib.jtvae_segment_mean(None, None, None, 0, 450, 452, None) == 0
    assert lib.jtvae_segment_ce(None, None, None, 0, None, None, None, None) == 0
    assert lib.jtvae_segment_ce(None, None, None, 3, None, None, None, None) == -1
    assert lib.jtvae_pack_batch(None, 0, None) == 0
    assert lib.jtvae_pack_batch(None, 3, None) == -1
    assert lib.jtvae_pack_batch(None, K.PACK_MAX + 1, None) == -2
    bad = (K.PackItem * 1)(K.PackItem(src=16, dst=16, ld_src=4, ld_dst=4, rows=3, cols=8, transpose=0))  # cols > ld
    assert lib.jtvae_pack_batch(bad, 1, None) == -2
    assert lib.jtvae_gemm_nt_pair(None, None, None) == -1
    p_ok = K.nt_problem(16, 452, 16, 452, 16, 452, 0, 450, 450)   # M = 0: nothing to do
    p_bad = K.nt_problem(16, 452, 16, 452, 16, 452, 8, 450, 450, mode=K.EPI_MUL_ACC)  # epilogue operands missing
    assert lib.jtvae_gemm_nt_pair(ctypes.byref(p_ok), ctypes.byref(p_bad), None) == -1
    assert lib.jtvae_gemm_nt_pair(ctypes.byref(p_ok), ctypes.byref(p_ok), None) == 0
    assert lib.jtvae_select_slab_rows(None, None, 0, None, 0, 450, 452, None) in (0, -1)
    assert lib.jtvae_row_dot(None, None, None, None, 0, 450, 452, None) == 0


def test_modules_refuse_cpu():
    import pytest
    import torch

    from fast_jtnn_hip import JTMPN, MPN, JTNNEncoder

    m = MPN(16, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 39), torch.zeros(3, 51), torch.zeros(2, 6).long(), torch.zeros(3, 6).long(), [(0, 2)])
    e = JTNNEncoder(16, 2, torch.nn.Embedding(5, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        e(torch.zeros(2).long(), torch.zeros(3).long(), torch.zeros(2, 1).long(), torch.zeros(3, 1).long(), [(0, 2)])
    j = JTMPN(16, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        j(torch.zeros(2, 38), torch.zeros(2, 44), torch.zeros(2, 15).long(), torch.zeros(2, 15).long(), [(0, 2)],
          torch.zeros(3, 16))


def test_state_dict_keys_match_reference_layout():
    """Names and shapes listed for checkpoint compatibility (reference model with vocab 385)."""
    from fast_jtnn_hip import JTNNVAE, synthetic

    m = JTNNVAE(synthetic.FakeVocab(385), 450, 56, 20, 3)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = {
        "mpn.W_i.weight": (450, 51), "mpn.W_h.weight": (450, 450), "mpn.W_o.weight": (450, 489),
        "mpn.W_o.bias": (450,), "jtnn.embedding.weight": (385, 450), "jtnn.outputNN.0.weight": (450, 900),
        "jtnn.outputNN.0.bias": (450,), "jtnn.GRU.W_z.weight": (450, 900), "jtnn.GRU.W_z.bias": (450,),
        "jtnn.GRU.W_r.weight": (450, 450), "jtnn.GRU.U_r.weight": (450, 450), "jtnn.GRU.U_r.bias": (450,),
        "jtnn.GRU.W_h.weight": (450, 900), "jtnn.GRU.W_h.bias": (450,), "jtmpn.W_i.weight": (450, 44),
        "jtmpn.W_h.weight": (450, 450), "jtmpn.W_o.weight": (450, 488), "jtmpn.W_o.bias": (450,),
        "A_assm.weight": (450, 28), "T_mean.weight": (28, 450), "G_var.bias": (28,),
    }
    for k, shp in want.items():
        assert sd.get(k) == shp, (k, sd.get(k))


def test_which_calls_give_up_the_interpreter_lock():
    """Single launches are bound through the handle that keeps the interpreter lock (microseconds each); the host-side
    preparation routines and the calls that issue a whole pass through the one that releases it, so that background
    preparation threads run while the stepping thread is inside them."""
    from fast_jtnn_hip import _capi as K

    keeps = ctypes.PyDLL(K.LIB_PATH).jtvae_abi_version._flags_
    releases = ctypes.CDLL(K.LIB_PATH).jtvae_abi_version._flags_
    assert keeps != releases
    for name in K.SIGNATURES:
        want = releases if name.startswith(K.HOST_PREFIXES) or name in K.LONG_CALLS else keeps
        assert getattr(K.lib, name)._flags_ == want, name
    assert set(K.LONG_CALLS) <= set(K.SIGNATURES)
    for name in ("jtvae_gemm_nt", "jtvae_csr_gather_sum", "jtvae_gru_update"):
        assert getattr(K.lib, name)._flags_ == keeps
    for name in ("jtvae_tree_encoder_backward", "jtvae_graph_mpn_forward", "jtvae_prep_transpose_csr"):
        assert getattr(K.lib, name)._flags_ == releases
