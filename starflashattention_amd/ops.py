"""Host-side operators over the C ABI: torch tensors in, HIP kernels underneath.

torch is plumbing here (device memory, current stream); every byte of compute happens in
libStarFlashAttention.so.  The two public operators:

  flash_decode(...)   same arguments, meaning and side effects as the reference's
                      star_flash_attn.mha_fwd_cuda (src/flash_api.cpp:42-68): mutates `o` and the
                      two caches in place and returns `o`.
  flash_attn_fwd(...) prefill forward (new entry point; the reference is decode-only).
"""
import ctypes
import math

import torch

from . import _lib

_DTYPES = {torch.float16: _lib.DTYPE_FP16, torch.bfloat16: _lib.DTYPE_BF16}

_workspaces = {}          # (device index, stream) -> torch.uint8 tensor
_sync_checks = False


def set_sync_checks(enabled: bool):
    """When on, flash_decode synchronises after every call and raises if a seq_len[b] was out of
    range (the kernel rejects such rows on the device either way)."""
    global _sync_checks
    _sync_checks = bool(enabled)


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require(cond, msg):
    if not cond:
        raise RuntimeError("star_flash_attn: " + msg)


def _check_gpu_tensor(t, name, dtype=None, shape=None, device=None):
    _require(isinstance(t, torch.Tensor), f"{name} must be a tensor")
    _require(t.is_cuda, f"{name} must live on a HIP device (got {t.device})")
    if device is not None:
        _require(t.device == device, f"{name} is on {t.device}, expected {device}")
    if dtype is not None:
        _require(t.dtype == dtype, f"{name} has dtype {t.dtype}, expected {dtype}")
    if shape is not None:
        _require(tuple(t.shape) == tuple(shape), f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    _require(t.is_contiguous(), f"{name} must be contiguous")


def _workspace(device, nbytes):
    lib = _lib.load()
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        grown = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        if ws is None:
            _lib.check(lib.sfa_decode_reset_status(ctypes.c_void_p(grown.data_ptr()), _stream_ptr(device)))
        else:
            grown[:256].copy_(ws[:256])      # the sticky status block survives growth (stream-ordered, no sync)
        _workspaces[key] = ws = grown
    return ws


def release_workspaces():
    """Drop the cached decode scratch of every (device, stream).  Pending sticky flags are lost:
    call check_decode_status() first if they matter."""
    _workspaces.clear()


def flash_decode(qkv, q_bias, k_bias, v_bias, k_cache_table, v_cache_table, seq_len, o,
                 batch_size, memory_max_len, num_heads, head_dim, rotary_embedding_dim,
                 max_input_length, num_layer, idx_layer, *, num_splits=0, _sized_by_query_heads=False,
                 rotary_cos_table=None, rotary_sin_table=None, softmax_scale=None, kv_layout="blmhd",
                 block_table=None, num_heads_kv=None):
    """One decode step (see include/star_flash_attn.h, sfa_decode).  Returns `o` (same tensor).
    kv_layout: "blmhd" = the reference's [B, L, M, H, D] caches; "blhmd" = head-major [B, L, H, M, D];
    "paged" = page pools [num_pages, L, page_size, H, D] addressed through block_table (int32
    [B, pages_per_seq]); memory_max_len is then the capacity of one sequence.
    num_heads_kv (grouped queries, an extension): qkv is then [B, num_heads + 2*num_heads_kv, D] (q heads,
    k heads, v heads), k_bias / v_bias and the caches carry num_heads_kv heads."""
    lib = _lib.load()
    _require(isinstance(qkv, torch.Tensor) and qkv.dtype in _DTYPES,
             f"qkv must be a float16 or bfloat16 tensor (got {getattr(qkv, 'dtype', type(qkv))})")
    dt, dev = qkv.dtype, qkv.device
    B, H, D, M, L = int(batch_size), int(num_heads), int(head_dim), int(memory_max_len), int(num_layer)
    Hkv = H if num_heads_kv is None else int(num_heads_kv)
    _require(Hkv > 0 and H % Hkv == 0, f"num_heads={H} must be a multiple of num_heads_kv={Hkv}")
    _check_gpu_tensor(qkv, "qkv", dt, (B, 3, H, D) if Hkv == H else (B, H + 2 * Hkv, D))
    _check_gpu_tensor(o, "o", dt, (B, H, D), dev)
    _check_gpu_tensor(seq_len, "seq_len", torch.int32, (B,), dev)
    _require(kv_layout in _lib.KV_LAYOUTS, f"kv_layout must be one of {sorted(_lib.KV_LAYOUTS)} (got {kv_layout!r})")
    if kv_layout == "paged":
        _require(isinstance(block_table, torch.Tensor) and block_table.dim() == 2,
                 "kv_layout='paged' needs block_table, an int32 [batch, pages_per_seq] tensor")
        _require(isinstance(k_cache_table, torch.Tensor) and k_cache_table.dim() == 5,
                 "k_cache_table must be a [num_pages, num_layer, page_size, num_heads, head_dim] pool")
        num_pages, page_size = int(k_cache_table.shape[0]), int(k_cache_table.shape[2])
        cache_shape = (num_pages, L, page_size, Hkv, D)
        _check_gpu_tensor(block_table, "block_table", torch.int32, (B, int(block_table.shape[1])), dev)
    else:
        _require(block_table is None, "block_table is only meaningful with kv_layout='paged'")
        cache_shape = (B, L, M, Hkv, D) if kv_layout == "blmhd" else (B, L, Hkv, M, D)
    _check_gpu_tensor(k_cache_table, "k_cache_table", dt, cache_shape, dev)
    _check_gpu_tensor(v_cache_table, "v_cache_table", dt, cache_shape, dev)
    biases = []
    for name, t in (("q_bias", q_bias), ("k_bias", k_bias), ("v_bias", v_bias)):
        if t is None or t.numel() == 0:
            biases.append(None)
        else:
            _check_gpu_tensor(t, name, dt, (H if name == "q_bias" else Hkv, D), dev)
            biases.append(t)
    for name, t in (("rotary_cos_table", rotary_cos_table), ("rotary_sin_table", rotary_sin_table)):
        if t is not None:
            _check_gpu_tensor(t, name, dt, (M, int(rotary_embedding_dim) // 2), dev)
    with torch.cuda.device(dev):
        # the library sizes the split count by the KV heads (one workgroup serves a whole group)
        S = int(num_splits) if num_splits and num_splits > 0 else lib.sfa_decode_auto_splits(B, Hkv, D, M)
        ws = _workspace(dev, lib.sfa_decode_workspace_bytes(B, H, D, M, S))
        if _sized_by_query_heads:
            # (tests) the older contract of the C ABI: a caller that leaves the split count to the library sizes its
            # workspace with sfa_decode_workspace_bytes(..., 0), which knows the query-head count only
            S = 0
            ws = torch.empty(lib.sfa_decode_workspace_bytes(B, H, D, M, 0), dtype=torch.uint8, device=dev)
            _lib.check(lib.sfa_decode_reset_status(ctypes.c_void_p(ws.data_ptr()), _stream_ptr(dev)))
        a = _lib.DecodeArgs()
        a.qkv = qkv.data_ptr()
        a.q_bias, a.k_bias, a.v_bias = (b.data_ptr() if b is not None else None for b in biases)
        a.o = o.data_ptr()
        a.seq_len = seq_len.data_ptr()
        a.k_cache_table = k_cache_table.data_ptr()
        a.v_cache_table = v_cache_table.data_ptr()
        a.rotary_cos_table = rotary_cos_table.data_ptr() if rotary_cos_table is not None else None
        a.rotary_sin_table = rotary_sin_table.data_ptr() if rotary_sin_table is not None else None
        a.batch_size, a.memory_max_len, a.num_heads, a.head_dim = B, M, H, D
        a.head_dim_inv = float(softmax_scale) if softmax_scale else 1.0 / math.sqrt(D)
        a.rotary_embedding_dim = int(rotary_embedding_dim)
        a.max_input_length = int(max_input_length)
        a.stride = (H + 2 * Hkv) * D
        a.num_heads_kv = Hkv
        a.num_layer, a.idx_layer = L, int(idx_layer)
        a.num_splits = S
        a.dtype = _DTYPES[dt]
        a.workspace = ws.data_ptr()
        a.workspace_bytes = ws.numel()
        a.kv_layout = _lib.KV_LAYOUTS[kv_layout]
        if kv_layout == "paged":
            a.page_size, a.num_pages = page_size, num_pages
            a.block_table = block_table.data_ptr()
            a.block_table_stride = int(block_table.shape[1])
        _lib.check(lib.sfa_decode(ctypes.byref(a), _stream_ptr(dev)))
        if _sync_checks:
            check_decode_status(dev)
    return o


def check_decode_status(device=None):
    """Synchronise the current stream and raise if any decode call on it saw a bad seq_len."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        return
    with torch.cuda.device(device):
        st = lib.sfa_decode_poll_status(ctypes.c_void_p(ws.data_ptr()), _stream_ptr(device))
        if st != _lib.SFA_OK:
            lib.sfa_decode_reset_status(ctypes.c_void_p(ws.data_ptr()), _stream_ptr(device))
        _lib.check(st)


def flash_attn_fwd(q, k, v, causal=False, softmax_scale=None, out=None, return_lse=False, fast_scale=False):
    """O = softmax(mask(Q K^T * scale)) V.   q [B,Hq,Sq,D], k/v [B,Hkv,Sk,D] (any batch/head/seq
    strides, D contiguous), fp16 or bf16, D in {64,128,256}.  causal is bottom-right aligned.
    fast_scale=True (ignored with return_lse) allows the prescaled-Q kernels: ~5 % faster, the scale is
    folded into Q in 16 bit, so the score error grows with the logits (include/star_flash_attn.h)."""
    lib = _lib.load()
    _require(isinstance(q, torch.Tensor) and q.dtype in _DTYPES,
             f"q must be float16 or bfloat16 (got {getattr(q, 'dtype', type(q))})")
    dt, dev = q.dtype, q.device
    for name, t in (("q", q), ("k", k), ("v", v)):
        _require(t.is_cuda and t.device == dev, f"{name} must be on {dev}")
        _require(t.dtype == dt, f"{name} dtype {t.dtype} != {dt}")
        _require(t.dim() == 4, f"{name} must be [batch, heads, seq, head_dim]")
        _require(t.stride(3) == 1, f"{name}: head_dim must be contiguous")
    B, Hq, Sq, D = q.shape
    _require(k.shape == v.shape and k.shape[0] == B and k.shape[3] == D,
             f"k/v shapes {tuple(k.shape)}/{tuple(v.shape)} do not match q {tuple(q.shape)}")
    Hkv, Sk = k.shape[1], k.shape[2]
    if out is None:
        out = torch.empty((B, Hq, Sq, D), dtype=dt, device=dev)
    else:
        _require(out.shape == q.shape and out.dtype == dt and out.device == dev and out.stride(3) == 1,
                 "out must match q in shape/dtype/device with contiguous head_dim")
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev) if return_lse else None
    a = _lib.PrefillArgs()
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.lse = lse.data_ptr() if lse is not None else None
    a.batch, a.heads_q, a.heads_kv, a.seqlen_q, a.seqlen_k, a.head_dim = B, Hq, Hkv, Sq, Sk, D
    for dst, t in ((a.q_stride, q), (a.k_stride, k), (a.v_stride, v), (a.o_stride, out)):
        dst[0], dst[1], dst[2] = t.stride(0), t.stride(1), t.stride(2)
    a.softmax_scale = float(softmax_scale) if softmax_scale else 0.0
    a.causal = 1 if causal else 0
    a.dtype = _DTYPES[dt]
    a.fast_scale = 1 if fast_scale else 0
    with torch.cuda.device(dev):
        _lib.check(lib.sfa_prefill_fwd(ctypes.byref(a), _stream_ptr(dev)))
    return (out, lse) if return_lse else out


def compute_rotary_table(max_seq_len, rot_dim, dtype=torch.float16, device="cuda"):
    lib = _lib.load()
    device = torch.device(device)
    cos = torch.empty((max_seq_len, rot_dim // 2), dtype=dtype, device=device)
    sin = torch.empty_like(cos)
    with torch.cuda.device(device):
        _lib.check(lib.sfa_compute_rotary_table(cos.data_ptr(), sin.data_ptr(), max_seq_len, rot_dim,
                                                _DTYPES[dtype], _stream_ptr(device)))
    return cos, sin


def fill_16bit(t, value):
    """t[...] = value for a contiguous fp16/bf16 tensor (init_half_array's job)."""
    lib = _lib.load()
    _require(t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES, "fill_16bit: contiguous 16-bit HIP tensor")
    bits = torch.tensor([value], dtype=t.dtype).view(torch.int16).item() & 0xFFFF
    with torch.cuda.device(t.device):
        _lib.check(lib.sfa_fill_16bit(t.data_ptr(), bits, t.numel(), _stream_ptr(t.device)))
    return t
