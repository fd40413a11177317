"""Oracle: single-token decode attention with fused RoPE + KV append.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference's ground truth, /root/reference/examples/python/
testFlashDecoder.py:
  * LlamaRotaryEmbedding.forward   py:18-26  (inv_freq fp32, angle = t*inv_freq,
                                              repeat_interleave(2))
  * rotate_half                    py:28-43  (interleaved pairs (-x1,x0,-x3,x2..))
  * apply_rotary_pos_emb           py:46-58  (x*cos + rotate_half(x)*sin, then
                                              cast back to the storage dtype)
  * LlamaAttention.forward         py:68-94  (cat(cache[:, layer, :cur-1], k_new),
                                              softmax(q.K^T/sqrt(D)).V)
and the kernel-side conventions the CUDA source intends
(/root/reference/src/flash_attn.cu):
  * seq_len[b] = tokens ALREADY cached; RoPE position = seq_len[b]
    (cu:570-571, cu:600, cu:685)  => oracle.current_seq_len == seq_len[b] + 1
  * k_rot and v_new are appended at cache[b, layer, seq_len[b]] (cu:696-701)
  * partial rotary: pair j rotates iff 2j < rot_dim, with
    inv_freq_j = base^(-2j/rot_dim) (cu:161-173,193,209)
  * optional q/k/v bias [H,D] added before RoPE (params.h:15-17, cu:662 TODO)
  * optional cos/sin LUT [M, rot_dim/2] in the storage dtype (params.h:27-29)
Defects of the CUDA kernel listed in SURVEY.md section 8(a) are NOT reproduced.
"""
import numpy as np

from .numerics import round_to


def _inv_freq(rot_dim, base=10000.0):
    # py:11  inv_freq = 1. / (base ** (arange(0, dim, 2).float() / dim))  (fp32)
    e = np.arange(0, rot_dim, 2, dtype=np.float32) / np.float32(rot_dim)
    return (np.float32(1.0) / np.power(np.float32(base), e)).astype(np.float32)


def rotary_table_ref(max_seq_len, rot_dim, dtype="fp16", base=10000.0):
    """cos/sin LUT [max_seq_len, rot_dim/2], rounded to the storage dtype.

    Intended semantics of compute_rotary_table<half> (cu:512-538; declared
    flash_attn.h:9-10), with the row stride fixed to rot_dim/2 (the reference
    strides rows by rot_dim and overruns its own allocation)."""
    t = np.arange(max_seq_len, dtype=np.float32)[:, None]
    ang = (t * _inv_freq(rot_dim, base)[None, :]).astype(np.float32)
    c = np.cos(ang.astype(np.float64))
    s = np.sin(ang.astype(np.float64))
    return round_to(c, dtype), round_to(s, dtype)


def rope_interleaved(x, pos, rot_dim, base=10000.0, cos=None, sin=None):
    """Rotate interleaved pairs (x[2j], x[2j+1]) of the last axis by pos*inv_freq_j.

    x: [..., D] float; returns float64 (unrounded).  cos/sin (each [rot_dim/2])
    override the on-the-fly trig when a LUT row is supplied."""
    x = np.asarray(x, dtype=np.float64)
    out = x.copy()
    if rot_dim <= 0:
        return out
    if cos is None:
        ang = (np.float32(pos) * _inv_freq(rot_dim, base)).astype(np.float32)
        cos = np.cos(ang.astype(np.float64))
        sin = np.sin(ang.astype(np.float64))
    cos = np.asarray(cos, dtype=np.float64)
    sin = np.asarray(sin, dtype=np.float64)
    xe = x[..., 0:rot_dim:2]
    xo = x[..., 1:rot_dim:2]
    out[..., 0:rot_dim:2] = xe * cos - xo * sin     # q*cos + (-x_odd)*sin
    out[..., 1:rot_dim:2] = xo * cos + xe * sin     # q*cos + ( x_even)*sin
    return out


def decode_ref(qkv, k_cache, v_cache, seq_len, idx_layer, rot_dim, dtype="fp16",
               q_bias=None, k_bias=None, v_bias=None,
               cos_table=None, sin_table=None,
               round_q=True, round_kv=True, scale=None, base=10000.0):
    """One decode step for the whole batch.

    qkv      [B,3,H,D]    float (values representable in `dtype`)
    k_cache  [B,L,M,H,D]  float, MUTATED at [b, idx_layer, seq_len[b]]
    v_cache  [B,L,M,H,D]  float, MUTATED likewise
    seq_len  [B] int      tokens already cached per sample (ragged allowed)
    returns dict(o=[B,H,D] float32 unrounded, o_rounded, k_row, v_row, q_rot)
    """
    qkv = np.asarray(qkv)
    B, three, H, D = qkv.shape
    assert three == 3
    M = k_cache.shape[2]
    if scale is None:
        scale = 1.0 / np.sqrt(float(D))       # flash_api.cpp:26 head_dim_inv
    rdt = dtype if round_kv else None
    o = np.zeros((B, H, D), dtype=np.float64)
    k_rows = np.zeros((B, H, D), dtype=np.float32)
    v_rows = np.zeros((B, H, D), dtype=np.float32)
    q_rots = np.zeros((B, H, D), dtype=np.float32)
    for b in range(B):
        pos = int(seq_len[b])
        if not (0 <= pos < M):
            # cc:141-142 pair (4096,4096) overruns the cache in the reference;
            # the replacement must reject it.
            raise ValueError(f"seq_len[{b}]={pos} outside [0, memory_max_len={M})")
        q = qkv[b, 0].astype(np.float64)
        k = qkv[b, 1].astype(np.float64)
        v = qkv[b, 2].astype(np.float64)
        if q_bias is not None:
            q = q + np.asarray(q_bias, dtype=np.float64)
        if k_bias is not None:
            k = k + np.asarray(k_bias, dtype=np.float64)
        if v_bias is not None:
            v = v + np.asarray(v_bias, dtype=np.float64)
        c = s = None
        if cos_table is not None and rot_dim > 0:
            c, s = cos_table[pos], sin_table[pos]
        q_rot = rope_interleaved(q, pos, rot_dim, base, c, s)
        k_rot = rope_interleaved(k, pos, rot_dim, base, c, s)
        q_rot = round_to(q_rot, dtype if round_q else None).astype(np.float64)
        k_rot = round_to(k_rot, rdt)
        v_new = round_to(v, rdt)
        k_cache[b, idx_layer, pos] = k_rot
        v_cache[b, idx_layer, pos] = v_new
        K = k_cache[b, idx_layer, :pos + 1].astype(np.float64)   # [T,H,D]
        V = v_cache[b, idx_layer, :pos + 1].astype(np.float64)
        sc = np.einsum("hd,thd->ht", q_rot, K) * scale
        sc -= sc.max(axis=1, keepdims=True)
        p = np.exp(sc)
        p /= p.sum(axis=1, keepdims=True)
        o[b] = np.einsum("ht,thd->hd", p, V)
        k_rows[b], v_rows[b], q_rots[b] = k_rot, v_new, q_rot
    return dict(o=o.astype(np.float32), o_rounded=round_to(o, dtype),
                k_row=k_rows, v_row=v_rows, q_rot=q_rots)
