"""Oracle: scaled-dot-product attention forward (the prefill path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference has no prefill kernel (flash_attn.h:3-11 is its whole surface);
the semantics are the SDPA definition its own oracle uses for one row
(testFlashDecoder.py:84-92: softmax(q.K^T/sqrt(D)).V), extended to Sq rows with
an optional causal mask.  sdpa_torch_cpu is the north_star's CPU baseline
(PyTorch eager F.scaled_dot_product_attention on the host cores) and doubles as
an independent cross-check of the numpy restatement.
"""
import numpy as np


def sdpa_ref(q, k, v, causal=False, scale=None, return_lse=False):
    """q [B,Hq,Sq,D], k/v [B,Hkv,Sk,D] float -> o [B,Hq,Sq,D] float32 (fp64 math).

    GQA: query head h reads kv head h // (Hq//Hkv).
    causal: key j visible to query i iff j <= i + (Sk - Sq) (bottom-right
    aligned; identical to torch's is_causal when Sq == Sk).  A query row with
    no visible key yields zeros (and lse = -inf)."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    B, Hq, Sq, D = q.shape
    Hkv, Sk = k.shape[1], k.shape[2]
    assert Hq % Hkv == 0
    g = Hq // Hkv
    if scale is None:
        scale = 1.0 / np.sqrt(float(D))
    o = np.zeros((B, Hq, Sq, D), dtype=np.float64)
    lse = np.full((B, Hq, Sq), -np.inf, dtype=np.float64)
    mask = None
    if causal:
        i = np.arange(Sq)[:, None]
        j = np.arange(Sk)[None, :]
        mask = j <= i + (Sk - Sq)
    for b in range(B):
        for h in range(Hq):
            s = (q[b, h] @ k[b, h // g].T) * scale
            if mask is not None:
                s = np.where(mask, s, -np.inf)
            m = s.max(axis=1, keepdims=True)
            m = np.where(np.isfinite(m), m, 0.0)
            p = np.exp(s - m)
            l = p.sum(axis=1, keepdims=True)
            safe = np.where(l > 0, l, 1.0)
            o[b, h] = (p @ v[b, h // g]) / safe
            with np.errstate(divide="ignore"):
                lse[b, h] = (m + np.log(l))[:, 0]
    if return_lse:
        return o.astype(np.float32), lse.astype(np.float32)
    return o.astype(np.float32)


def sdpa_torch_cpu(q, k, v, causal=False, scale=None):
    """PyTorch eager SDPA on the host CPU in fp32 (torch tensors in, tensor out)."""
    import torch
    import torch.nn.functional as F
    q, k, v = (t.detach().to("cpu", torch.float32) for t in (q, k, v))
    if k.shape[1] != q.shape[1]:
        g = q.shape[1] // k.shape[1]
        k = k.repeat_interleave(g, dim=1)
        v = v.repeat_interleave(g, dim=1)
    return F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
