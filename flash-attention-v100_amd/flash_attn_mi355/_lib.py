"""ctypes binding of libfa_mi355.so (the C ABI in include/fa_mi355.h).

The library is the product: there is NO CPU / PyTorch fallback.  If it is missing or does
not match the header this module raises at import time."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FA_MI355_LIB") or os.path.join(_HERE, "libfa_mi355.so")   # env: A/B experiment builds

FA_FP16, FA_BF16, FA_FP8_E4M3 = 0, 1, 2
FA_ABI_VERSION = 3
FA_FLAG_KEEP_WINDOW = 1
FA_FLAG_NO_DKV_SPLIT = 2
FA_FLAG_DS_HANDOFF = 4
FA_FLAG_FWD_KEY_SPLIT = 8

_i64, _i32, _f32, _u64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_uint64
_ptr = ctypes.c_void_p


class FaParams(ctypes.Structure):
    """Mirror of `struct fa_params` - field order MUST match include/fa_mi355.h
    (checked against the header by tests/test_abi.py and against sizeof at load)."""
    _fields_ = [
        ("q", _ptr), ("k", _ptr), ("v", _ptr), ("o", _ptr), ("lse", _ptr),
        ("q_batch_stride", _i64), ("q_row_stride", _i64), ("q_head_stride", _i64),
        ("k_batch_stride", _i64), ("k_row_stride", _i64), ("k_head_stride", _i64),
        ("v_batch_stride", _i64), ("v_row_stride", _i64), ("v_head_stride", _i64),
        ("o_batch_stride", _i64), ("o_row_stride", _i64), ("o_head_stride", _i64),
        ("lse_batch_stride", _i64), ("lse_head_stride", _i64),
        ("dout", _ptr), ("dq", _ptr), ("dk", _ptr), ("dv", _ptr), ("softmax_d", _ptr),
        ("do_batch_stride", _i64), ("do_row_stride", _i64), ("do_head_stride", _i64),
        ("dq_batch_stride", _i64), ("dq_row_stride", _i64), ("dq_head_stride", _i64),
        ("dk_batch_stride", _i64), ("dk_row_stride", _i64), ("dk_head_stride", _i64),
        ("dv_batch_stride", _i64), ("dv_row_stride", _i64), ("dv_head_stride", _i64),
        ("batch", _i32), ("nheads_q", _i32), ("nheads_k", _i32), ("seqlen_q", _i32),
        ("seqlen_k", _i32), ("head_dim", _i32), ("dtype", _i32), ("kv_dtype", _i32),
        ("softmax_scale", _f32), ("softcap", _f32),
        ("is_causal", _i32), ("window_left", _i32), ("window_right", _i32),
        ("alibi_slopes", _ptr), ("alibi_batch_stride", _i64),
        ("p_dropout", _f32), ("philox_seed", _u64), ("philox_offset", _u64), ("dmask", _ptr),
        ("cu_seqlens_q", _ptr), ("cu_seqlens_k", _ptr), ("seqused_k", _ptr),
        ("total_q", _i32), ("total_k", _i32),
        ("block_table", _ptr), ("block_table_batch_stride", _i64),
        ("page_block_size", _i32), ("head_dim_v", _i32),
        ("cache_seqlens", _ptr), ("cache_batch_idx", _ptr), ("cache_leftpad", _ptr),
        ("k_new", _ptr), ("v_new", _ptr),
        ("knew_batch_stride", _i64), ("knew_row_stride", _i64), ("knew_head_stride", _i64),
        ("vnew_batch_stride", _i64), ("vnew_row_stride", _i64), ("vnew_head_stride", _i64),
        ("seqlen_new", _i32), ("rotary_dim", _i32),
        ("rotary_cos", _ptr), ("rotary_sin", _ptr),
        ("rotary_interleaved", _i32), ("seqlen_ro", _i32),
        ("k_descale", _f32), ("v_descale", _f32),
        ("num_splits", _i32), ("flags", _i32),
        ("workspace", _ptr), ("workspace_bytes", ctypes.c_size_t),
    ]


EXPORTS = ["fa_abi_version", "fa_params_size", "fa_last_error", "fa_build_info",
           "fa_fwd_workspace_bytes", "fa_bwd_workspace_bytes", "fa_fwd_kvcache_workspace_bytes",
           "fa_fwd", "fa_bwd", "fa_varlen_fwd", "fa_varlen_bwd", "fa_fwd_kvcache",
           "fa_gather_rows", "fa_scatter_rows"]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or flash-attention-v100_amd/build.py). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"libfa_mi355.so does not export {name}")
    lib.fa_abi_version.restype = ctypes.c_int
    lib.fa_params_size.restype = ctypes.c_size_t
    lib.fa_last_error.restype = ctypes.c_char_p
    lib.fa_build_info.restype = ctypes.c_char_p
    for name in ("fa_fwd_workspace_bytes", "fa_bwd_workspace_bytes", "fa_fwd_kvcache_workspace_bytes"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.POINTER(FaParams)]
    for name in ("fa_fwd", "fa_bwd", "fa_varlen_fwd", "fa_varlen_bwd", "fa_fwd_kvcache"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(FaParams), ctypes.c_void_p]
    i64 = ctypes.c_int64
    lib.fa_gather_rows.restype = ctypes.c_int
    lib.fa_gather_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, i64, i64, i64, ctypes.c_void_p]
    lib.fa_scatter_rows.restype = ctypes.c_int
    lib.fa_scatter_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, i64, i64, ctypes.c_int, ctypes.c_void_p]
    if lib.fa_abi_version() != FA_ABI_VERSION:
        raise ImportError(f"libfa_mi355.so ABI {lib.fa_abi_version()} != binding {FA_ABI_VERSION}")
    if lib.fa_params_size() != ctypes.sizeof(FaParams):
        raise ImportError(f"fa_params size mismatch: library {lib.fa_params_size()} vs ctypes "
                          f"{ctypes.sizeof(FaParams)}")
    return lib


lib = _load()


def call(name, params, stream):
    """Invoke an op; raise RuntimeError (like TORCH_CHECK -> RuntimeError in the reference,
    kernel/fused_mha_api.cpp) with the library's message on failure."""
    rc = getattr(lib, name)(ctypes.byref(params), ctypes.c_void_p(stream))
    if rc != 0:
        msg = lib.fa_last_error().decode(errors="replace")
        raise RuntimeError(f"{name} failed ({rc}): {msg}")


def call_rows(name, *args):
    """fa_gather_rows / fa_scatter_rows (plain-argument entry points)."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.fa_last_error().decode(errors='replace')}")
