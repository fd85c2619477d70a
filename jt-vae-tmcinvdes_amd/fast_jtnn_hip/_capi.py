"""ctypes binding of libjtvae_hip.so (declared in include/jtvae_hip.h).

There is no fallback: if the library is missing or a symbol is absent the import
fails, and every call raises on a non-zero return code.  PyTorch is used only to
own device memory and to name the stream; tensors cross the boundary as raw
device pointers, so CALLERS MUST KEEP EVERY TENSOR ALIVE UNTIL THE LAUNCH THAT
READS IT HAS BEEN ENQUEUED (a temporary freed right after ``.data_ptr()`` can be
handed to the next allocation and overwritten first).

``KernelTimer`` is the measurement hook: inside ``with KernelTimer() as t:`` every
call is bracketed by HIP events on the launch stream and tagged with its
algorithmic work (bytes for the row kernels, flops for the GEMMs), which is what
bench.py turns into achieved GB/s and TFLOP/s per kernel.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# JTVAE_HIP_LIB: another build of the library (same ABI), e.g. to compare two kernel versions in one session
LIB_PATH = os.environ.get("JTVAE_HIP_LIB") or os.path.join(_HERE, "libjtvae_hip.so")

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
EPI_PLAIN, EPI_GRU_BLEND, EPI_MUL_ACC = 0, 1, 2
ABI_VERSION = 19


class Epilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p),
        ("add", C.c_void_p),
        ("ld_add", C.c_int32),
        ("add_idx", C.c_void_p),
        ("act", C.c_int32),
        ("pre_out", C.c_void_p),
        ("ld_pre", C.c_int32),
        ("mode", C.c_int32),
        ("aux0", C.c_void_p),
        ("aux1", C.c_void_p),
        ("out2", C.c_void_p),
        ("ld_aux", C.c_int32),
        ("a_rows", C.c_void_p),
        ("c_rows", C.c_void_p),
        ("zero_row", C.c_int32),
        ("aux_sf", C.c_void_p),
        ("aux_t", C.c_int32),
        ("aux_slab_stride", C.c_size_t),
    ]


class NtProblem(C.Structure):
    _fields_ = [
        ("A", C.c_void_p),
        ("lda", C.c_int32),
        ("W", C.c_void_p),
        ("ldw", C.c_int32),
        ("C", C.c_void_p),
        ("ldc", C.c_int32),
        ("M", C.c_int32),
        ("N", C.c_int32),
        ("K", C.c_int32),
        ("epi", C.POINTER(Epilogue)),
    ]


class TreeFwd(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("h", "uh", "s", "g", "t", "z", "p", "sf", "Pz", "Pr", "Ph", "wid", "rowptr",
                                            "col", "Wzh", "Whh", "Ur", "bu")] +
                [(n, C.c_int32) for n in ("H", "ld", "pad_row", "n_levels")] +
                [("f_rows", C.POINTER(C.c_void_p)), ("f_n", C.POINTER(C.c_int32)),
                 ("u_rows", C.POINTER(C.c_void_p)), ("u_n", C.POINTER(C.c_int32))])


class TreeBwd(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("h", "uh", "s", "t", "z", "p", "sf", "active", "Pr", "wid", "trowptr",
                                            "tcol", "WzhT", "WhhT", "UrT", "dh", "dz", "dp", "du", "ds", "dg",
                                            "dhdir", "a_r")] +
                [(n, C.c_int32) for n in ("T", "H", "ld", "pad_row", "n_levels")] +
                [("e_rows", C.POINTER(C.c_void_p)), ("e_n", C.POINTER(C.c_int32)), ("n_dep", C.POINTER(C.c_int32)),
                 ("off_e", C.POINTER(C.c_int64)), ("off_j", C.POINTER(C.c_int64))])


class Csr(C.Structure):  # jtvae_csr_t
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("n_rows", C.c_int32)]


class TreeEncFwd(C.Structure):  # jtvae_tree_enc_fwd_t
    _fields_ = ([(n, C.c_int32) for n in ("NM", "B", "NW", "V", "H", "ld", "crowded", "train")] +
                [("E", C.c_void_p), ("word", Csr), ("root", Csr)] +
                [(n, C.c_void_p) for n in ("Wzx", "Whx", "Wox", "Wr", "Woh", "bz", "bh", "bout", "wid_root")] +
                [("sweep", C.POINTER(TreeFwd))] +
                [(n, C.c_void_p) for n in ("Eu", "Po", "nsum", "tv")])


class TreeEncBwd(C.Structure):  # jtvae_tree_enc_bwd_t
    _fields_ = ([(n, C.c_int32) for n in ("NM", "B", "NW", "V", "H", "ld", "crowded")] +
                [(n, C.c_void_p) for n in ("Eu", "nsum", "tv", "dtv", "dmess")] + [("ld_dmess", C.c_int32)] +
                [(n, C.c_void_p) for n in ("WohT", "WzxT", "WhxT", "WoxT", "WrT")] +
                [(n, Csr) for n in ("ident", "root_t", "word_t", "o_pieces", "o_key", "zh_pieces", "zh_key",
                                    "r_pieces", "r_key")] +
                [("sweep", C.POINTER(TreeBwd)), ("n_e", C.c_int32)] +
                [(n, C.c_void_p) for n in ("ss", "gs", "hs", "zx", "zy", "ux", "uy")] +
                [("n_z", C.c_int32), ("n_u", C.c_int32)] +
                [(n, C.c_void_p) for n in ("dpre_root", "dnsum", "part_side", "part_main", "dPo", "dPz", "dPh", "dPr",
                                           "dEu", "dE_p")] +
                [("ws", C.c_void_p * 8), ("ws_bytes", C.c_size_t * 8)] +
                [(n, C.c_void_p) for n in ("dWout", "dWz", "dWh", "dWr", "dUr", "dbout", "dbz", "dbh", "dbu", "dE")])


class PackItem(C.Structure):
    _fields_ = [
        ("src", C.c_void_p),
        ("dst", C.c_void_p),
        ("ld_src", C.c_int32),
        ("ld_dst", C.c_int32),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("transpose", C.c_int32),
    ]




class MpnFwdArgs(C.Structure):  # jtvae_mpn_fwd_t
    _fields_ = ([(n, C.c_void_p) for n in ("fbonds", "fatoms", "Wi", "Wh", "Woa", "Woh", "bo", "tree", "b_rowptr",
                                           "b_col", "a_rowptr", "a_col", "seg", "msg", "nei", "anei", "ah", "out")] +
                [(n, C.c_int32) for n in ("ld_fb", "ld_fa", "ld_wi", "ld_woa", "n_tree", "n_bonds", "n_atoms", "n_seg",
                                          "fb_dim", "fa_dim", "H", "ld", "depth")])


class MpnBwdArgs(C.Structure):  # jtvae_mpn_bwd_t
    _fields_ = ([(n, C.c_void_p) for n in ("dout", "fbonds", "fatoms", "WhT", "WohT", "msg", "nei", "anei", "ah",
                                           "at_rowptr", "at_col", "bt_rowptr", "bt_col", "seg", "seg_of", "dpre_o",
                                           "danei", "dpre", "dnei", "dbinput", "dW_i", "dW_h", "dW_o", "db_o", "dtree",
                                           "workspace")] +
                [("workspace_bytes", C.c_size_t)] +
                [(n, C.c_int32) for n in ("ld_fb", "ld_fa", "ld_dwi", "ld_dwo", "n_tree", "n_bonds", "n_atoms",
                                          "n_seg", "fb_dim", "fa_dim", "H", "ld", "depth")])


class GraphSetC(C.Structure):
    _fields_ = [("n_graphs", C.c_int64), ("atom_ptr", C.c_void_p), ("bond_ptr", C.c_void_p), ("atom_feats", C.c_void_p),
                ("bonds", C.c_void_p), ("bond_feats", C.c_void_p), ("atom_nid", C.c_void_p), ("atom_fdim", C.c_int32),
                ("bond_fdim", C.c_int32)]


class MessIndexC(C.Structure):
    _fields_ = [("n_mess", C.c_int64), ("n_nodes", C.c_int64), ("src", C.c_void_p), ("dst", C.c_void_p),
                ("node_base", C.c_void_p)]


class GraphTablesC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fatoms", "fbonds", "agraph", "bgraph", "scope")]


class GraphCsrC(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("fatoms", "fbonds", "a_rowptr", "a_col", "b_rowptr", "b_col")] +
                [("a_cap", C.c_int64), ("b_cap", C.c_int64)] +
                [(n, C.c_void_p) for n in ("at_rowptr", "at_col", "bt_rowptr", "bt_col", "seg", "seg_of")] +
                [("nnz", C.c_int64 * 4)])


class TreeSetC(C.Structure):
    _fields_ = [("n_trees", C.c_int64)] + [(n, C.c_void_p) for n in ("node_ptr", "wid", "nbr_ptr", "nbr", "cand_ptr",
                                                                      "label")]


class TreeTablesC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fnode", "fmess", "node_graph", "mess_graph", "scope")] + \
               [("Kn", C.c_int32), ("Km", C.c_int32)]


class TreeCsrC(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("wid", "src", "dst", "n_rowptr", "n_col", "m_rowptr", "m_col")] +
                [("m_cap", C.c_int64), ("sf", C.c_void_p), ("roots", C.c_void_p), ("nnz", C.c_int64),
                 ("max_sf", C.c_int32)])


PACK_MAX = 16
E_NULL, E_SHAPE, E_ALIGN, E_RANGE, E_WORKSPACE = -1, -2, -3, -4, -5
_i32, _vp, _sz, _i64 = C.c_int32, C.c_void_p, C.c_size_t, C.c_int64

# name -> (restype, argtypes); must list every function of include/jtvae_hip.h
SIGNATURES = {
    "jtvae_abi_version": (C.c_int, []),
    "jtvae_error_string": (C.c_char_p, [C.c_int]),
    "jtvae_gemm_nt": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp]),
    "jtvae_gemm_nt_pair": (C.c_int, [C.POINTER(NtProblem), C.POINTER(NtProblem), _vp]),
    "jtvae_gemm_tn_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "jtvae_gemm_tn": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _sz,
                                _vp]),
    "jtvae_gemm_tn_static_max_bytes": (_sz, []),
    "jtvae_gemm_tn_static": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "jtvae_pack_batch": (C.c_int, [C.POINTER(PackItem), _i32, _vp]),
    "jtvae_segment_ce": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "jtvae_prep_padded_to_csr": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _i64, C.POINTER(C.c_int64)]),
    "jtvae_prep_transpose_csr": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "jtvae_prep_stable_from": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "jtvae_nt_launch_count": (C.c_uint64, []),
    "jtvae_set_split_products": (C.c_int, [C.c_int]),
    "jtvae_set_small_tile_rows": (C.c_int, [C.c_int]),
    "jtvae_tree_gru_forward": (C.c_int, [_vp, _vp]),
    "jtvae_tree_gru_backward": (C.c_int, [_vp, _vp]),
    "jtvae_tree_encoder_backward": (C.c_int, [_vp, _vp, _vp]),
    "jtvae_tree_encoder_forward": (C.c_int, [_vp, _vp, _vp]),
    "jtvae_graph_mpn_forward": (C.c_int, [_vp, _vp]),
    "jtvae_graph_mpn_backward": (C.c_int, [_vp, _vp]),
    "jtvae_prep_group_by_key": (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "jtvae_prep_features_to_u8": (C.c_int, [_vp, _i64, _vp, C.POINTER(C.c_int32)]),
    "jtvae_widen_u8_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_prep_backward_once": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jtvae_prep_forward_once": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jtvae_prep_decoder_plan": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp]),
    "jtvae_copy_rows": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_csr_gather_sum": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    "jtvae_segment_mean": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_segment_mean_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_gru_iter0": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "jtvae_gru_gate_gather": (C.c_int, [_vp, _vp, _sz, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                        _i32, _i32, _vp]),
    "jtvae_gru_update": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32,
                                   _i32, _vp]),
    "jtvae_gru_bwd_pointwise": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32,
                                          _i32, _vp]),
    "jtvae_gru_bwd_dh": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _i32, _vp, _vp, _vp, _vp,
                                   _i32, _i32, _i32, _vp]),
    "jtvae_gru_bwd_gather": (C.c_int, [_vp, _vp, _sz, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                       _vp, _vp, _i32, _i32, _vp]),
    "jtvae_select_slab_rows": (C.c_int, [_vp, _vp, _sz, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_row_dot": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "jtvae_row_dot_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "jtvae_onehot_linear": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "jtvae_onehot_linear_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "jtvae_onehot_linear_bwd": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _sz, _vp]),
    "jtvae_tensorize_graph_counts": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "jtvae_tensorize_graphs": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32]),
    "jtvae_tensorize_tree_counts": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "jtvae_tensorize_trees": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32]),
    "jtvae_select_candidates_counts": (C.c_int, [_vp, _vp, _i64, _vp]),
    "jtvae_select_candidates": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
}


HOST_PREFIXES = ("jtvae_prep_", "jtvae_tensorize_", "jtvae_select_")  # host-only routines, run for milliseconds
# whole passes issued by one call (a dozen to ~70 launches, 0.05-0.3 ms of host time)
LONG_CALLS = ("jtvae_tree_encoder_forward", "jtvae_tree_encoder_backward", "jtvae_graph_mpn_forward",
              "jtvae_graph_mpn_backward", "jtvae_tree_gru_forward", "jtvae_tree_gru_backward")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libjtvae_hip.so is not built (expected at %s); run `make -C jt-vae-tmcinvdes_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH
        )
    # Two handles on the one library.  Single kernel launches return in microseconds: they are called WITHOUT giving
    # up the interpreter lock (PyDLL), because a stepping thread that drops it ~300 times per step has to win it
    # back each time from a prefetch thread that may be in the middle of Python code.  The host-side index
    # routines (jtvae_prep_*) run for milliseconds and do release it (CDLL), so prefetch threads overlap them; so
    # do the calls that issue a whole pass (LONG_CALLS): the ~0.6 ms per step the stepping thread spends inside
    # them is when the preparation threads get their turns without stalling it (prefetched fresh-holder rate at
    # 1024 molecules per step: 199 000-206 000 molecules/s against 164 000-192 000 with the lock held).
    host, launch = C.CDLL(LIB_PATH), C.PyDLL(LIB_PATH)

    class _Lib:
        pass
    lib = _Lib()
    for name, (res, args) in SIGNATURES.items():
        releases = name.startswith(HOST_PREFIXES) or name in LONG_CALLS
        fn = getattr(host if releases else launch, name)  # AttributeError if not exported
        fn.restype = res
        fn.argtypes = args
        setattr(lib, name, fn)
    got = lib.jtvae_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libjtvae_hip.so ABI %d, host code expects %d" % (got, ABI_VERSION))
    return lib


lib = _load()


class JtvaeError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise JtvaeError("%s failed: %s (code %d)" % (what, lib.jtvae_error_string(rc).decode(), rc))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of PyTorch's current stream on the current device (0 = the null stream).
    Called once per launch, so it takes the direct binding when this PyTorch has it."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t, offset_elems: int = 0):
    """Device pointer of a tensor (None -> NULL), optionally offset by elements."""
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()


LD_ALIGN = 4


def round_ld(h: int) -> int:
    return (h + LD_ALIGN - 1) // LD_ALIGN * LD_ALIGN


def rows(n: int, ld: int, device, zero: bool = False) -> torch.Tensor:
    """A padded row buffer [n, ld] fp32."""
    f = torch.zeros if zero else torch.empty
    return f((max(n, 0), ld), dtype=torch.float32, device=device)


# ---- measurement hook ------------------------------------------------------------------------


class KernelTimer:
    """Collects (kernel, algorithmic bytes, algorithmic flops, HIP-event pair) per C-ABI call."""

    active = None

    def __init__(self, streams=False):
        self.records = []
        self.streams = [] if streams else None  # stream handle of every record, for timelines

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def summary(self):
        """kernel -> dict(launches, ms, bytes, flops); synchronises once."""
        torch.cuda.synchronize()
        out = OrderedDict()
        for name, nbytes, flops, e0, e1, _tag, n in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0, flops=0))
            d["launches"] += n
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out

    def detail(self):
        """(kernel, shape tag) -> dict(launches, ms, bytes, flops)."""
        torch.cuda.synchronize()
        out = OrderedDict()
        for name, nbytes, flops, e0, e1, tag, n in self.records:
            d = out.setdefault((name, tag), dict(launches=0, ms=0.0, bytes=0, flops=0))
            d["launches"] += n
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out


_SYNC_EACH = bool(os.environ.get("JTVAE_SYNC_EACH"))  # debugging aid: a stall or fault then names its launch


def _run(name, fn, args, nbytes=0, flops=0, tag=""):
    t = KernelTimer.active
    if t is None:
        check(fn(*args), name)
        if _SYNC_EACH:
            torch.cuda.current_stream().synchronize()
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    before = lib.jtvae_nt_launch_count() if name == "gemm_nt" else 0
    e0.record()
    rc = fn(*args)
    e1.record()
    # a pair / fused update that had to run as two launches counts as two
    n = max(1, lib.jtvae_nt_launch_count() - before) if name == "gemm_nt" else 1
    t.records.append((name, int(nbytes), int(flops), e0, e1, tag, n))
    if t.streams is not None:
        t.streams.append(torch.cuda.current_stream().cuda_stream)
    check(rc, name)


# ---- thin call wrappers (sizes are taken from explicit arguments, never guessed) -----------
# Tensor arguments are raw device pointers (ints) or None.  `nnz` arguments exist only to price
# the call for KernelTimer (rows read = nnz); None leaves the gather unpriced.


def gemm_nt(A, lda, W, ldw, Cout, ldc, M, N, K, *, bias=None, add=None, ld_add=0, add_idx=None,
            act=ACT_NONE, pre_out=None, ld_pre=0, mode=EPI_PLAIN, aux0=None, aux1=None, out2=None,
            ld_aux=0, a_rows=None, c_rows=None, zero_row=-1, aux_sf=None, aux_t=0, aux_slab_stride=0):
    e = Epilogue(bias, add, ld_add, add_idx, act, pre_out, ld_pre, mode, aux0, aux1, out2, ld_aux,
                 a_rows, c_rows, zero_row, aux_sf, aux_t, aux_slab_stride)
    # an addend looked up through add_idx is a small per-word table (L2-resident): not priced per row
    extra = (1 if add and not add_idx else 0) + (1 if pre_out else 0) + {EPI_PLAIN: 0, EPI_GRU_BLEND: 3,
                                                                          EPI_MUL_ACC: 3}[mode]
    _run("gemm_nt", lib.jtvae_gemm_nt, (A, lda, W, ldw, Cout, ldc, M, N, K, C.byref(e), stream_ptr()),
         nbytes=4 * (M * K + N * K + M * N * (1 + extra)), flops=2 * M * N * K,
         tag="M=%d N=%d K=%d mode=%d" % (M, N, K, mode))


class small_tile_rows:
    """Context: products issued by this thread take the one-tile-per-block kernels up to ``rows`` rows only (0: never)."""

    def __init__(self, rows):
        self.rows = rows

    def __enter__(self):
        self.before = lib.jtvae_set_small_tile_rows(self.rows) if self.rows is not None else None

    def __exit__(self, *exc):
        if self.rows is not None:
            lib.jtvae_set_small_tile_rows(self.before)


def set_split_products(on: bool) -> bool:
    """Large products in the split-bf16 form (default) or on the fp32 matrix instruction; returns the previous
    setting.  Both are fp32-accurate; they differ in the last bits."""
    return bool(lib.jtvae_set_split_products(1 if on else 0))


def nt_problem(A, lda, W, ldw, Cout, ldc, M, N, K, **epi):
    """Arguments of one gemm_nt call, for gemm_nt_pair (keyword arguments as in gemm_nt)."""
    names = ("bias", "add", "ld_add", "add_idx", "act", "pre_out", "ld_pre", "mode", "aux0", "aux1", "out2", "ld_aux",
             "a_rows", "c_rows", "zero_row", "aux_sf", "aux_t", "aux_slab_stride")
    defaults = dict(bias=None, add=None, ld_add=0, add_idx=None, act=ACT_NONE, pre_out=None, ld_pre=0, mode=EPI_PLAIN,
                    aux0=None, aux1=None, out2=None, ld_aux=0, a_rows=None, c_rows=None, zero_row=-1, aux_sf=None,
                    aux_t=0, aux_slab_stride=0)
    unknown = set(epi) - set(names)
    if unknown:
        raise TypeError("unknown epilogue arguments %s" % sorted(unknown))
    defaults.update(epi)
    e = Epilogue(*[defaults[n] for n in names])
    p = NtProblem(A, lda, W, ldw, Cout, ldc, M, N, K, C.pointer(e))
    p._keep = e  # the problem points at its epilogue
    return p


def gemm_nt_pair(p0, p1):
    """Two independent products (see nt_problem) in one launch where the library supports the pair."""
    def price(p):
        e = p._keep
        extra = (1 if e.add and not e.add_idx else 0) + (1 if e.pre_out else 0) + (3 if e.mode != EPI_PLAIN else 0)
        return 4 * (p.M * p.K + p.N * p.K + p.M * p.N * (1 + extra)), 2 * p.M * p.N * p.K
    (b0, f0), (b1, f1) = price(p0), price(p1)
    _run("gemm_nt", lib.jtvae_gemm_nt_pair, (C.byref(p0), C.byref(p1), stream_ptr()), nbytes=b0 + b1, flops=f0 + f1,
         tag="pair M=%d+%d N=%d K=%d" % (p0.M, p1.M, p0.N, p0.K))


def gemm_tn(X, ldx, Y, ldy, dW, lddw, M, N, K, accumulate, device, x_rows=None, y_rows=None, dbias=None):
    """dW (+)= X^T Y over the M rows (or row pairs); dbias (+)= column sums of the same X rows."""
    nbytes = lib.jtvae_gemm_tn_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    _run("gemm_tn", lib.jtvae_gemm_tn, (X, ldx, x_rows, Y, ldy, y_rows, dW, lddw, dbias, M, N, K, int(accumulate),
                                        ws.data_ptr(), nbytes, stream_ptr()),
         nbytes=4 * (M * N + M * K + N * K), flops=2 * M * N * K, tag="M=%d N=%d K=%d" % (M, N, K))


def pack_batch(items):
    """items: list of (src_ptr, ld_src, dst_ptr, ld_dst, rows, cols, transpose); one launch per 16."""
    for i in range(0, len(items), PACK_MAX):
        chunk = items[i:i + PACK_MAX]
        arr = (PackItem * len(chunk))(*[PackItem(src=it[0], ld_src=it[1], dst=it[2], ld_dst=it[3], rows=it[4],
                                                cols=it[5], transpose=it[6]) for it in chunk])
        _run("pack_batch", lib.jtvae_pack_batch, (arr, len(chunk), stream_ptr()),
             nbytes=sum(8 * it[4] * it[5] for it in chunk))


def copy_rows(src, ld_src, dst, ld_dst, nrows, ncols):
    _run("copy_rows", lib.jtvae_copy_rows, (src, ld_src, dst, ld_dst, nrows, ncols, stream_ptr()),
         nbytes=8 * nrows * ncols)


def widen_u8_rows(src, dst, ld_dst, nrows, ncols):
    _run("widen_u8_rows", lib.jtvae_widen_u8_rows, (src, dst, ld_dst, nrows, ncols, stream_ptr()),
         nbytes=nrows * (ncols + 4 * ld_dst))


ONEHOT_LDS_MAX = 150 << 10


def onehot_fits(F: int, ld: int) -> bool:
    """The transposed weight of a feature layer fits the LDS of a CU (see jtvae_onehot_linear)."""
    return F * ld * 4 + 128 * ((F + 3) // 4) * 4 <= ONEHOT_LDS_MAX and ld // 4 <= 256


def onehot_linear(feats, n_rows, F, Wt, ldw, bias, out, ld, H):
    _run("onehot_linear", lib.jtvae_onehot_linear, (feats, n_rows, F, Wt, ldw, bias, out, ld, H, stream_ptr()),
         nbytes=n_rows * (F + 4 * H) + 4 * F * H)


def onehot_linear_bwd(feats, n_rows, F, dOut, ld, H, dW, lddw, accumulate, device):
    nbytes = lib.jtvae_onehot_linear_bwd_workspace_bytes(n_rows, F, ld)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    _run("onehot_linear_bwd", lib.jtvae_onehot_linear_bwd,
         (feats, n_rows, F, dOut, ld, H, dW, lddw, int(accumulate), ws.data_ptr(), nbytes, stream_ptr()),
         nbytes=n_rows * (F + 4 * H) + 4 * F * H)


def csr_gather_sum(out, src0, n_src0, src1, rowptr, col, n_rows, H, ld, relu_mask=None, accumulate=False,
                   acc2=None, nnz=None, relu_src=False):
    row_b = 4 * H
    extra = (1 if relu_mask else 0) + (1 if accumulate else 0) + (2 if acc2 else 0)
    nbytes = 0 if nnz is None else (nnz + n_rows * (1 + extra)) * row_b + 4 * (n_rows + 1 + nnz)
    _run("csr_gather_sum", lib.jtvae_csr_gather_sum,
         (out, src0, n_src0, src1, rowptr, col, n_rows, H, ld, relu_mask, int(accumulate), acc2, int(relu_src),
          stream_ptr()),
         nbytes=nbytes)


def segment_mean(out, rows_, seg, n_seg, H, ld, n_rows=None):
    _run("segment_mean", lib.jtvae_segment_mean, (out, rows_, seg, n_seg, H, ld, stream_ptr()),
         nbytes=0 if n_rows is None else 4 * H * (n_rows + n_seg) + 8 * n_seg)


def segment_mean_bwd(drows, dout, act, seg, seg_of, n_rows, H, ld, n_seg=0):
    _run("segment_mean_bwd", lib.jtvae_segment_mean_bwd,
         (drows, dout, act, seg, seg_of, n_rows, H, ld, stream_ptr()),
         nbytes=4 * H * (2 * n_rows + n_seg) + 4 * n_rows)


def gru_iter0(Pz, Ph, wid, z, p, h1, n_mess, H, ld, pad_row=0):
    _run("gru_iter0", lib.jtvae_gru_iter0, (Pz, Ph, wid, z, p, h1, n_mess, H, ld, pad_row, stream_ptr()),
         nbytes=4 * H * n_mess * (3 if z else 1) + 4 * n_mess)  # rows written; the two word tables are L2 hits


def gru_gate_gather(h, Uh, slab_stride, t_iter, sf, Pr, wid, rowptr, col, rows, n_rows, s, g, t, H, ld, nnz=None):
    # rows: h and Uh per edge, s / g (/ t) written (the W_r x row comes from the per-word table, an L2 hit)
    nbytes = 0 if nnz is None else 4 * H * (2 * nnz + n_rows * (3 if t else 2)) + 4 * (3 * n_rows + nnz)
    _run("gru_gate_gather", lib.jtvae_gru_gate_gather,
         (h, Uh, slab_stride, t_iter, sf, Pr, wid, rowptr, col, rows, n_rows, s, g, t, H, ld, stream_ptr()),
         nbytes=nbytes)


def gru_update(S, G, ld, Wz, Wh, ldw, Pz, Ph, ldp, wid, rows, n_rows, Z, P, Hout, H, zero_row=0):
    """Both products of a GRU level and the blend in one launch; priced as the two gemm_nt calls it replaces."""
    _run("gemm_nt", lib.jtvae_gru_update, (S, G, ld, Wz, Wh, ldw, Pz, Ph, ldp, wid, rows, n_rows, Z, P, Hout, H,
                                           zero_row, stream_ptr()),
         nbytes=4 * (2 * n_rows * H + 2 * H * H + 4 * n_rows * H), flops=4 * n_rows * H * H,
         tag="gru_update M=%d N=%d K=%d" % (n_rows, H, H))


def sweeps_enabled():
    """The one-call level sweeps are used unless every launch is being timed or synchronised on its own."""
    return KernelTimer.active is None and not _SYNC_EACH


def row_list_arrays(lists):
    """Host arrays (device pointers, counts) of a sequence of RowList-like objects (``ptr()``, ``n``)."""
    ptrs = (C.c_void_p * len(lists))(*[r.ptr() for r in lists])
    counts = (C.c_int32 * len(lists))(*[int(r.n) for r in lists])
    return ptrs, counts


def tree_gru_forward(args):
    check(lib.jtvae_tree_gru_forward(C.byref(args), stream_ptr()), "tree_gru_forward")


def tree_gru_backward(args):
    check(lib.jtvae_tree_gru_backward(C.byref(args), stream_ptr()), "tree_gru_backward")


def tree_encoder_forward(args, side=None):
    """``side``: raw handle of the helper stream, or None."""
    check(lib.jtvae_tree_encoder_forward(C.byref(args), stream_ptr(), side), "tree_encoder_forward")


def tree_encoder_backward(args, side=None):
    """``side``: raw handle of the helper stream, or None."""
    check(lib.jtvae_tree_encoder_backward(C.byref(args), stream_ptr(), side), "tree_encoder_backward")


def csr_arg(csr, n_rows=None):
    """jtvae_csr_t of an uploaded CSR (``rowptr`` / ``col`` device tensors)."""
    return Csr(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.n_rows if n_rows is None else n_rows)


def pointer_array(tensors):
    """Host array of the device addresses of a sequence of tensors."""
    return (C.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])


def graph_mpn_forward(args):
    check(lib.jtvae_graph_mpn_forward(C.byref(args), stream_ptr()), "graph_mpn_forward")


def graph_mpn_backward(args):
    check(lib.jtvae_graph_mpn_backward(C.byref(args), stream_ptr()), "graph_mpn_backward")


def gemm_tn_workspace_bytes(M, N, Kd):
    return int(lib.jtvae_gemm_tn_workspace_bytes(M, N, Kd))


def gru_bwd_pointwise(dh, z, p, s, slab_stride, t_iter, sf, rows, n_rows, dzpre, dppre, ds, H, ld, pad_row=0):
    _run("gru_bwd_pointwise", lib.jtvae_gru_bwd_pointwise,
         (dh, z, p, s, slab_stride, t_iter, sf, rows, n_rows, dzpre, dppre, ds, H, ld, pad_row, stream_ptr()),
         nbytes=4 * H * n_rows * 7 + 8 * n_rows)


def gru_bwd_dh(du, UrT, ldw, dhdir, dh, rows, n_rows, z, p, s, slab_stride, t_iter, sf, dzpre, dppre, ds, H, ld,
               pad_row=0):
    """dh[rows] = du . UrT^T + dhdir[rows] and, from that value, the pre-activation gradients of the level: the
    product that follows gru_bwd_gather with gru_bwd_pointwise in its epilogue."""
    _run("gemm_nt", lib.jtvae_gru_bwd_dh,
         (du, UrT, ldw, dhdir, dh, rows, n_rows, z, p, s, slab_stride, t_iter, sf, dzpre, dppre, ds, H, ld, pad_row,
          stream_ptr()),
         nbytes=4 * (H * H + n_rows * H * 9), flops=2 * n_rows * H * H, tag="bwd_dh M=%d N=%d K=%d" % (n_rows, H, H))


def gru_bwd_gather(h, Uh, slab_stride, t_iter, sf, active, Pr, wid, dg, ds, trowptr, tcol, rows, n_rows, dUh,
                   dhdir, H, ld, nnz=None, dh_ext=None):
    # rows: h, Uh (dh_ext) read and dUh, dhdir written per message; dg, ds read per edge (the table row of the
    # reader's word is an L2 hit and is not priced)
    nbytes = 0 if nnz is None else 4 * H * ((5 if dh_ext else 4) * n_rows + 2 * nnz) + 4 * (2 * n_rows + 2 * nnz)
    _run("gru_bwd_gather", lib.jtvae_gru_bwd_gather,
         (h, Uh, slab_stride, t_iter, sf, active, Pr, wid, dg, ds, trowptr, tcol, rows, n_rows, dh_ext, dUh, dhdir,
          H, ld, stream_ptr()), nbytes=nbytes)


def select_slab_rows(out, base, slab_stride, sel, n_rows, H, ld):
    _run("select_slab_rows", lib.jtvae_select_slab_rows, (out, base, slab_stride, sel, n_rows, H, ld, stream_ptr()),
         nbytes=8 * H * n_rows + 4 * n_rows)


def segment_ce(scores, seg_ptr, label_pos, n_seg, loss_seg, hit_seg, dscores, n_scores=0):
    _run("segment_ce", lib.jtvae_segment_ce, (scores, seg_ptr, label_pos, n_seg, loss_seg, hit_seg, dscores,
                                              stream_ptr()), nbytes=8 * n_scores + 16 * n_seg)


def row_dot(score, q, owner, cand, n_cand, H, ld, n_owner=None):
    # an owner's row is shared by its consecutive candidates: priced once per owner, not once per candidate
    _run("row_dot", lib.jtvae_row_dot, (score, q, owner, cand, n_cand, H, ld, stream_ptr()),
         nbytes=4 * H * (n_cand + (n_cand if n_owner is None else n_owner)) + 8 * n_cand)


def row_dot_bwd(dscore, q, owner, cand, own_ptr, n_cand, n_owner, dq, dcand, H, ld):
    _run("row_dot_bwd", lib.jtvae_row_dot_bwd, (dscore, q, owner, cand, own_ptr, n_cand, n_owner, dq, dcand, H, ld,
                                                stream_ptr()), nbytes=4 * H * (2 * n_cand + 2 * n_owner) + 12 * n_cand)


# ---- host-side index preparation (numpy arrays in host memory; no GPU involved) -----------------


def _np_ptr(a):
    return a.__array_interface__["data"][0] if a.size else None


def prep_padded_to_csr(table, n_src):
    """Zero-padded int64 table [rows, K] -> (rowptr int32 [rows+1], col int32 [nnz]), slot order kept."""
    import numpy as np
    table = np.ascontiguousarray(table, dtype=np.int64)
    if table.ndim != 2:
        raise ValueError("neighbour table must be 2-D, got shape %s" % (table.shape,))
    rows, k = table.shape
    rowptr = np.empty(rows + 1, dtype=np.int32)
    col = np.empty(max(rows * k, 1), dtype=np.int32)
    nnz = C.c_int64(0)
    rc = lib.jtvae_prep_padded_to_csr(_np_ptr(table), rows, k, int(n_src), _np_ptr(rowptr), _np_ptr(col),
                                      col.size, C.byref(nnz))
    if rc == E_RANGE:
        raise IndexError("neighbour index out of range [0, %d)" % n_src)
    check(rc, "prep_padded_to_csr")
    return rowptr, col[:nnz.value].copy()


def prep_transpose_csr(rowptr, col, n_src):
    import numpy as np
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    trowptr = np.empty(int(n_src) + 1, dtype=np.int32)
    tcol = np.empty(max(col.size, 1), dtype=np.int32)
    rc = lib.jtvae_prep_transpose_csr(_np_ptr(rowptr), _np_ptr(col), rowptr.size - 1, int(n_src),
                                      _np_ptr(trowptr), _np_ptr(tcol))
    if rc == E_RANGE:
        raise IndexError("neighbour index out of range [0, %d)" % n_src)
    check(rc, "prep_transpose_csr")
    return trowptr, tcol[:col.size]


def prep_stable_from(rowptr, col, T):
    import numpy as np
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = rowptr.size - 1
    sf = np.empty(max(n, 1), dtype=np.int32)
    check(lib.jtvae_prep_stable_from(_np_ptr(rowptr), _np_ptr(col), n, int(T), _np_ptr(sf)), "prep_stable_from")
    return sf[:n]


def prep_features_to_u8(x, out) -> bool:
    """Float rows -> bytes in ``out`` (same element count); False when some value is not a whole number in
    [0, 255] (``out`` is then incomplete)."""
    exact = C.c_int32(0)
    check(lib.jtvae_prep_features_to_u8(_np_ptr(x), x.size, _np_ptr(out), C.byref(exact)), "prep_features_to_u8")
    return bool(exact.value)


def prep_group_by_key(keys, n_keys):
    import numpy as np
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    keyptr = np.empty(int(n_keys) + 1, dtype=np.int32)
    order = np.empty(max(keys.size, 1), dtype=np.int32)
    rc = lib.jtvae_prep_group_by_key(_np_ptr(keys), keys.size, int(n_keys), _np_ptr(keyptr), _np_ptr(order))
    if rc == E_RANGE:
        raise IndexError("key out of range [0, %d)" % n_keys)
    check(rc, "prep_group_by_key")
    return keyptr, order[:keys.size]


def prep_backward_once(rowptr, col, seeds=None):
    """Backward plan of the compute-once schedule (see jtvae_prep_backward_once): dict with act (bool [n]),
    order, lev_off, n_dep, deg_sum, tdeg_sum (one entry per level)."""
    import numpy as np
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = rowptr.size - 1
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    act = np.empty(max(n, 1), np.uint8)
    order = np.empty(max(n, 1), np.int32)
    lev_off = np.empty(n + 2, np.int32)
    n_dep = np.empty(n + 1, np.int32)
    deg_sum = np.empty(n + 1, np.int64)
    tdeg_sum = np.empty(n + 1, np.int64)
    counts = np.zeros(2, np.int64)
    rc = lib.jtvae_prep_backward_once(_np_ptr(rowptr), _np_ptr(col), n, None if seeds is None else _np_ptr(seeds),
                                      -1 if seeds is None else seeds.size, _np_ptr(act), _np_ptr(order),
                                      _np_ptr(lev_off), _np_ptr(n_dep), _np_ptr(deg_sum), _np_ptr(tdeg_sum),
                                      _np_ptr(counts))
    if rc == E_RANGE:
        raise IndexError("message dependencies are out of range or cyclic")
    check(rc, "prep_backward_once")
    n_act, n_lev = int(counts[0]), int(counts[1])
    return dict(act=act[:n].view(np.bool_), order=order[:n_act], lev_off=lev_off[:n_lev + 1], n_dep=n_dep[:n_lev],
                deg_sum=deg_sum[:n_lev], tdeg_sum=tdeg_sum[:n_lev])


def prep_forward_once(rowptr, col, sf_true, T):
    """Forward plan of the compute-once schedule (see jtvae_prep_forward_once): (sf, order, keyptr, u_rows, u_ptr,
    nnz_ptr)."""
    import numpy as np
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    sf_true = np.ascontiguousarray(sf_true, dtype=np.int32)
    n = rowptr.size - 1
    nk = max(int(T), 1) + 2
    sf, order, u_rows = (np.empty(max(n, 1), np.int32) for _ in range(3))
    keyptr, u_ptr, nnz_ptr = np.empty(nk, np.int32), np.empty(nk, np.int32), np.empty(nk, np.int64)
    rc = lib.jtvae_prep_forward_once(_np_ptr(rowptr), _np_ptr(col), n, _np_ptr(sf_true), int(T), _np_ptr(sf),
                                     _np_ptr(order), _np_ptr(keyptr), _np_ptr(u_rows), _np_ptr(u_ptr), _np_ptr(nnz_ptr))
    if rc == E_RANGE:
        raise IndexError("message dependencies out of range")
    check(rc, "prep_forward_once")
    return sf[:n], order[:n], keyptr, u_rows[:int(u_ptr[-1])], u_ptr, nnz_ptr


def prep_decoder_plan(src, dst, wid, roots):
    """Decoder trace tables from the junction-tree edges (see jtvae_prep_decoder_plan); returns a dict of arrays."""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    wid = np.ascontiguousarray(wid, dtype=np.int32)
    roots = np.ascontiguousarray(roots, dtype=np.int32)
    n_mess, n_nodes, B = src.size, wid.size, roots.size
    n_dir = n_mess - 1
    deg = np.bincount(src[1:], minlength=max(n_nodes, 1)).astype(np.int64) if n_dir else np.zeros(1, np.int64)
    cap = int((deg * deg).sum()) + n_dir + 1
    rows = n_dir + B
    a = dict(dep_rp=np.empty(n_mess + 1, np.int32), dep_col=np.empty(cap, np.int32),
             stop_rp=np.empty(rows + 1, np.int32), stop_col=np.empty(cap, np.int32),
             stop_word=np.empty(max(rows, 1), np.int32), stop_ctx=np.empty(max(rows, 1), np.int32),
             stop_target=np.empty(max(rows, 1), np.float32), pred_mess=np.empty(max(n_dir, 1), np.int32),
             pred_ctx=np.empty(max(rows, 1), np.int32), pred_target=np.empty(max(rows, 1), np.int32))
    counts = np.zeros(4, np.int64)
    ptr_of = lambda x: x.__array_interface__["data"][0]  # noqa: E731
    rc = lib.jtvae_prep_decoder_plan(ptr_of(src), ptr_of(dst), n_mess, ptr_of(wid), n_nodes, ptr_of(roots), B,
                                     ptr_of(a["dep_rp"]), ptr_of(a["dep_col"]), cap, ptr_of(a["stop_rp"]),
                                     ptr_of(a["stop_col"]), cap, ptr_of(a["stop_word"]), ptr_of(a["stop_ctx"]),
                                     ptr_of(a["stop_target"]), ptr_of(a["pred_mess"]), ptr_of(a["pred_ctx"]),
                                     ptr_of(a["pred_target"]), ptr_of(counts))
    if rc == E_RANGE:
        raise IndexError("junction-tree tables do not describe trees (index out of range, cycle or missing "
                         "reverse edge)")
    check(rc, "prep_decoder_plan")
    n_steps, n_down, nnz_dep, nnz_stop = (int(v) for v in counts)
    n_stop = n_steps + B
    return dict(dep_rp=a["dep_rp"], dep_col=a["dep_col"][:nnz_dep], stop_rp=a["stop_rp"][:n_stop + 1],
                stop_col=a["stop_col"][:nnz_stop], stop_word=a["stop_word"][:n_stop], stop_ctx=a["stop_ctx"][:n_stop],
                stop_target=a["stop_target"][:n_stop], pred_mess=a["pred_mess"][:n_down],
                pred_ctx=a["pred_ctx"][:B + n_down], pred_target=a["pred_target"][:B + n_down])
