"""16-bit float helpers for the oracle (test infrastructure only).

numpy has float16 but no bfloat16; bf16 values travel as uint16 bit patterns
or as float32 arrays whose low 16 mantissa bits are zero.
"""
import numpy as np


def bf16_round(x):
    """Round float array to the nearest bf16 (ties to even); returns float32.

    NaN stays NaN (the integer trick alone would not guarantee it, see
    MI355X_MICROARCH.md "Correctness boundaries")."""
    x32 = np.asarray(x, dtype=np.float32)
    u = x32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    return np.where(np.isnan(x32), np.float32(np.nan), out).astype(np.float32)


def fp16_round(x):
    """Round to the nearest IEEE half; returns float32."""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def round_to(x, dtype):
    if dtype in ("fp16", "f16", "half"):
        return fp16_round(x)
    if dtype in ("bf16", "bfloat16"):
        return bf16_round(x)
    if dtype in ("fp32", "f32", None):
        return np.asarray(x, dtype=np.float32)
    raise ValueError(f"unknown dtype {dtype!r}")


def to_bits16(x, dtype):
    """float array (already representable) -> uint16 storage bits."""
    if dtype in ("fp16", "f16", "half"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)
    if dtype in ("bf16", "bfloat16"):
        r = bf16_round(x)
        return (r.view(np.uint32) >> 16).astype(np.uint16)
    raise ValueError(dtype)


def from_bits16(bits, dtype):
    bits = np.asarray(bits, dtype=np.uint16)
    if dtype in ("fp16", "f16", "half"):
        return bits.view(np.float16).astype(np.float32)
    if dtype in ("bf16", "bfloat16"):
        return (bits.astype(np.uint32) << 16).view(np.float32)
    raise ValueError(dtype)
