"""ctypes binding of libStarFlashAttention.so's C ABI (include/star_flash_attn.h).

There is NO fallback: if the HIP library is missing this raises, loudly, with the build command.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFA_LIB_PATH: load another build of the same ABI instead (the A/B library with the earlier kernel
# generations, libStarFlashAttention_ab.so -- pytest -m variants, tools/prefill_ab.py)
LIB_PATH = os.environ.get("SFA_LIB_PATH") or os.path.join(_HERE, "lib", "libStarFlashAttention.so")

SFA_OK = 0
ABI_VERSION = 4          # SFA_ABI_VERSION in include/star_flash_attn.h
SFA_ERR_SEQ_LEN_RANGE = -7
SFA_ERR_BLOCK_TABLE_RANGE = -8
DTYPE_FP16, DTYPE_BF16 = 0, 1

# every symbol include/star_flash_attn.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "sfa_abi_version", "sfa_status_string", "sfa_last_error",
    "sfa_decode_workspace_bytes", "sfa_decode_workspace_bytes_gqa", "sfa_decode_auto_splits", "sfa_decode_reset_status",
    "sfa_decode_poll_status", "sfa_decode", "sfa_prefill_fwd",
    "sfa_compute_rotary_table", "sfa_fill_16bit", "sfa_debug_set", "sfa_debug_get",
]


class DecodeArgs(ctypes.Structure):
    """struct sfa_decode_args"""
    _fields_ = [
        ("qkv", ctypes.c_void_p), ("q_bias", ctypes.c_void_p), ("k_bias", ctypes.c_void_p),
        ("v_bias", ctypes.c_void_p), ("o", ctypes.c_void_p), ("seq_len", ctypes.c_void_p),
        ("k_cache_table", ctypes.c_void_p), ("v_cache_table", ctypes.c_void_p),
        ("rotary_cos_table", ctypes.c_void_p), ("rotary_sin_table", ctypes.c_void_p),
        ("batch_size", ctypes.c_int), ("memory_max_len", ctypes.c_int), ("num_heads", ctypes.c_int),
        ("head_dim", ctypes.c_int), ("head_dim_inv", ctypes.c_float),
        ("rotary_embedding_dim", ctypes.c_int), ("max_input_length", ctypes.c_int),
        ("stride", ctypes.c_int), ("num_layer", ctypes.c_int), ("idx_layer", ctypes.c_int),
        ("num_splits", ctypes.c_int), ("dtype", ctypes.c_int),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("kv_layout", ctypes.c_int), ("page_size", ctypes.c_int), ("block_table", ctypes.c_void_p),
        ("block_table_stride", ctypes.c_int), ("num_pages", ctypes.c_int), ("num_heads_kv", ctypes.c_int),
    ]


KV_LAYOUTS = {"blmhd": 0, "blhmd": 1, "paged": 2}      # enum sfa_kv_layout


class PrefillArgs(ctypes.Structure):
    """struct sfa_prefill_args"""
    _fields_ = [
        ("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p), ("o", ctypes.c_void_p),
        ("lse", ctypes.c_void_p),
        ("batch", ctypes.c_int), ("heads_q", ctypes.c_int), ("heads_kv", ctypes.c_int),
        ("seqlen_q", ctypes.c_int), ("seqlen_k", ctypes.c_int), ("head_dim", ctypes.c_int),
        ("q_stride", ctypes.c_int64 * 3), ("k_stride", ctypes.c_int64 * 3),
        ("v_stride", ctypes.c_int64 * 3), ("o_stride", ctypes.c_int64 * 3),
        ("softmax_scale", ctypes.c_float), ("causal", ctypes.c_int), ("dtype", ctypes.c_int),
        ("fast_scale", ctypes.c_int),
    ]


class SfaError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"star_flash_attn: {message} (status {status})")
        self.status = status


_lib = None


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP kernels are not built and there is no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`python starflashattention_amd/build.py`) with ROCm's hipcc on PATH.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.sfa_abi_version.restype = ctypes.c_int
    lib.sfa_status_string.restype = ctypes.c_char_p
    lib.sfa_status_string.argtypes = [ctypes.c_int]
    lib.sfa_last_error.restype = ctypes.c_char_p
    lib.sfa_decode_workspace_bytes.restype = ctypes.c_size_t
    lib.sfa_decode_workspace_bytes.argtypes = [ctypes.c_int] * 5
    lib.sfa_decode_workspace_bytes_gqa.restype = ctypes.c_size_t
    lib.sfa_decode_workspace_bytes_gqa.argtypes = [ctypes.c_int] * 6
    lib.sfa_decode_auto_splits.restype = ctypes.c_int
    lib.sfa_decode_auto_splits.argtypes = [ctypes.c_int] * 4
    lib.sfa_decode_reset_status.restype = ctypes.c_int
    lib.sfa_decode_reset_status.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sfa_decode_poll_status.restype = ctypes.c_int
    lib.sfa_decode_poll_status.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sfa_decode.restype = ctypes.c_int
    lib.sfa_decode.argtypes = [ctypes.POINTER(DecodeArgs), ctypes.c_void_p]
    lib.sfa_prefill_fwd.restype = ctypes.c_int
    lib.sfa_prefill_fwd.argtypes = [ctypes.POINTER(PrefillArgs), ctypes.c_void_p]
    lib.sfa_compute_rotary_table.restype = ctypes.c_int
    lib.sfa_compute_rotary_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sfa_fill_16bit.restype = ctypes.c_int
    lib.sfa_fill_16bit.argtypes = [ctypes.c_void_p, ctypes.c_uint16, ctypes.c_size_t, ctypes.c_void_p]
    lib.sfa_debug_set.restype = ctypes.c_int
    lib.sfa_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
    if lib.sfa_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.sfa_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


PREFILL_KERNEL_NAMES = {1: "prefill_kernel<8 waves, 256 rows, exact>", 3: "prefill_kernel<8 waves, 256 rows, prescaled>",
                        10: "prefill_kernel<8 waves, 256 rows, exact>", 21: "prefill_kernel_bm128<prescaled>",
                        22: "prefill_kernel_bm128<exact>", 41: "prefill_w4_kernel<prescaled>", 42: "prefill_w4_kernel<exact>",
                        60: "prefill_w4d_kernel<head_dim 256>", 61: "prefill_d256_kernel"}


def debug_get(knob):
    """sfa_debug_get: a knob's value, or "last_prefill_kernel" (the kernel the last prefill call launched)."""
    lib = load()
    lib.sfa_debug_get.restype = ctypes.c_int
    lib.sfa_debug_get.argtypes = [ctypes.c_char_p]
    return lib.sfa_debug_get(knob.encode())


def last_prefill_kernel():
    k = debug_get("last_prefill_kernel")
    return PREFILL_KERNEL_NAMES.get(k, f"prefill_impl {k}")


def debug_set(knob, value):
    """Test / A-B hook (sfa_debug_set): pick a kernel variant; -1 = the library's own choice."""
    check(load().sfa_debug_set(knob.encode(), int(value)))


def check(status):
    if status != SFA_OK:
        raise SfaError(status, load().sfa_last_error().decode("utf-8", "replace"))
