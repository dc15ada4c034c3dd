"""ctypes binding of libsrfrd_hip.so (the C ABI declared in include/srfrd_hip.h).

There is no CPU fallback: if the library is missing this module raises, and every op above it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRFRD_LIB_PATH") or os.path.join(_HERE, "lib", "libsrfrd_hip.so")

MAX_BLOCKS = 8
MAX_D = 64
KINDS = {"SASRec": 0, "SRFR": 1, "SRFRN": 2, "SRFU_B": 3, "SRFU_F": 4, "SRFU_R": 5}
_ERR = {-1: "SRFRD_E_ARG (bad argument)", -2: "SRFRD_E_UNSUPPORTED (configuration outside the fused kernels: "
        "hidden width > 64, hidden width not divisible by num_heads, a debug tap of a kernel without taps, or a caller-provided "
        "scratch buffer too small for this sequence length)", -3: "SRFRD_E_DEVICE"}


class BlockOff(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "ln2_w", "ln2_b",
                                         "c1_w", "c1_b", "c2_w", "c2_b")]


class Layout(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("kind", "n_items", "max_len", "d_item", "d_fake", "D", "d_out", "n_labels",
                                          "n_blocks", "n_heads", "side_rows", "side_cols", "table_bf16", "reserved0")]
                + [("off_pos", C.c_int64), ("off_side", C.c_int64), ("blk", BlockOff * MAX_BLOCKS),
                   ("off_lc_w", C.c_int64), ("off_lc_b", C.c_int64), ("off_ll_w", C.c_int64), ("off_ll_b", C.c_int64),
                   ("n_dense", C.c_int64), ("n_table", C.c_int64)])


_P = C.c_void_p
_i, _i64, _u32, _d = C.c_int, C.c_int64, C.c_uint32, C.c_double
_LP = C.POINTER(Layout)

# name -> (restype, argtypes); must list every symbol include/srfrd_hip.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "srfrd_layout_init": (_i, [_LP, _i, _i, _i, _i, _i, _i, _i, _i]),
    "srfrd_lds_bytes": (_i, [_LP, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "srfrd_scratch_floats": (_i, [_LP, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "srfrd_bwd_grid": (_i, [_LP, _i, _i]),
    "srfrd_debug_shape": (_i, [_LP, _i, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "srfrd_packed_floats": (_i64, [_LP]),
    "srfrd_pack_weights": (_i, [_LP, _P, _P, _P, _d, _d, _d, _P]),
    "srfrd_encoder_fwd": (_i, [_LP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _d, _u32, _P, _i64,
                               _P, _P, _P, _P, _P, _P, _P, _P, _i64, _P, _i, _P]),
    "srfrd_encoder_fwd_last": (_i, [_LP, _P, _P, _P, _P, _P, _i, _i, _P, _P, _i64, _P]),
    "srfrd_encoder_bwd": (_i, [_LP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _d, _u32, _P, _i64,
                               _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P, _P, _P, _P, _i64, _P, _i, _P]),
    "srfrd_sched_ints": (_i64, [_i]),
    "srfrd_seq_order": (_i, [_P, _i, _i, _i, _P, _P]),
    "srfrd_encoder_fwd_sched": (_i, [_LP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _d, _u32, _P, _i64,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _i64, _P, _i, _P]),
    "srfrd_encoder_bwd_sched": (_i, [_LP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _d, _u32, _P, _i64,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P, _P, _P, _P, _i64, _P, _i, _P]),
    "srfrd_table_reduce": (_i, [_P, _P, _P, _i64, _i, _P, _P]),
    "srfrd_aux_floats": (_i64, [_LP, _i, _i]),
    "srfrd_reduce_dense": (_i, [_P, _i, _i64, _P, _P, _i, _P, _P, _P]),
    "srfrd_loss_stats": (_i, [_P, _i, _P, _P, _P]),
    "srfrd_step_begin": (_i, [_P, _d, _d, _d, _P]),
    "srfrd_adam_step": (_i, [_P, _P, _P, _P, _i64, _i64, _i64, _i64, _d, _d, _d, _P, _P, _P, _i64, _P]),
    "srfrd_adam_pack_step": (_i, [_LP, _P, _P, _P, _P, _i64, _i64, _i64, _d, _d, _d, _d, _P, _P, _P, _P, _P]),
    "srfrd_table_to_bf16": (_i, [_P, _i64, _P, _P]),
    "srfrd_loss_finalize": (_i, [_P, _P, _P]),
    "srfrd_l2_norms": (_i, [_P, _P, _P, _i, _i64, _d, _P, _P, _P, _P]),
    "srfrd_l2_apply": (_i, [_P, _P, _i64, _i64, _i64, _P, _P, _P, _P]),
    "srfrd_user_labels": (_i, [_i, _P, _i, _i, _P, _P]),
    "srfrd_check_ids": (_i, [_P, _P, _P, _P, _P, _P, _i64, _i64, _i64, _P, _P]),
    "srfrd_predict_logits": (_i, [_LP, _P, _P, _P, _i, _i, _P, _i, _i64, _P, _P, _P]),
    "srfrd_topk_workspace_bytes": (_i64, [_i, _i, _i64]),
    "srfrd_logits_topk": (_i, [_LP, _P, _P, _P, _i, _i, _i64, _i64, _i, _P, _i, _P, _P, _P, _P]),
    "srfrd_topk_merge": (_i, [_P, _P, _i, _i, _i, _P, _P, _P]),
    "srfrd_eval_rank": (_i, [_P, _i, _i, _P, _P, _P]),
    "srfrd_sample_batch": (_i, [_P, _P, _P, _i, _i, _i, _i, _u32, _u32, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the SRFRD HIP kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{what}: hipError_t {rc}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def make_layout(kind: str, n_items: int, max_len: int, d_item: int, d_fake: int = 0, n_labels: int = 0,
                n_blocks: int = 2, n_heads: int = 1) -> Layout:
    lay = Layout()
    check(lib().srfrd_layout_init(C.byref(lay), KINDS[kind], n_items, max_len, d_item, d_fake, n_labels, n_blocks,
                                  n_heads), "srfrd_layout_init")
    return lay


def lds_bytes(lay: Layout, L: int):
    f, b = _i64(0), _i64(0)
    check(lib().srfrd_lds_bytes(C.byref(lay), L, C.byref(f), C.byref(b)), "srfrd_lds_bytes")
    return f.value, b.value


def scratch_floats(lay: Layout, B: int, L: int):
    f, b = _i64(0), _i64(0)
    check(lib().srfrd_scratch_floats(C.byref(lay), B, L, C.byref(f), C.byref(b)), "srfrd_scratch_floats")
    return f.value, b.value
