"""star_flash_attn for MI355X (gfx950): fused decode attention + prefill forward.

Thin host layer over libStarFlashAttention.so (HIP kernels behind the C ABI in
include/star_flash_attn.h).  No CPU fallback: importing the operators without the built
library raises.
"""
from .ops import (  # noqa: F401
    flash_decode, flash_attn_fwd, compute_rotary_table, fill_16bit,
    check_decode_status, set_sync_checks, release_workspaces,
)
from ._lib import SfaError, LIB_PATH, debug_set, debug_get, last_prefill_kernel  # noqa: F401

__version__ = "0.1.0"
