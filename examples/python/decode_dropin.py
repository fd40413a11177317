#!/usr/bin/env python3
"""Drop-in check of the `star_flash_attn` extension on PyTorch-ROCm.

Calls star_flash_attn.mha_fwd_cuda exactly the way the reference's example does
(reference examples/python/testFlashDecoder.py:124-126, positional, fp16, B=1 H=32 D=128, 4 layers)
and compares with a plain PyTorch computation of the same decode step on the GPU."""
import math

import torch
import star_flash_attn


def torch_decode(qkv, k_cache, v_cache, cur, layer):
    """softmax(rope(q) . [K_cache ; rope(k)]^T / sqrt(D)) . [V_cache ; v] with cur = tokens incl. the new one."""
    B, _, H, D = qkv.shape
    q, k, v = (qkv[:, i].float() for i in range(3))
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=qkv.device).float() / D))
    ang = (cur - 1) * inv
    cos, sin = ang.cos().repeat_interleave(2), ang.sin().repeat_interleave(2)

    def rot(x):
        y = torch.empty_like(x)
        y[..., 0::2], y[..., 1::2] = -x[..., 1::2], x[..., 0::2]
        return (x * cos + y * sin).to(qkv.dtype).float()

    q, k = rot(q), rot(k)
    K = torch.cat([k_cache[:, layer, :cur - 1].float(), k[:, None]], 1)     # [B,T,H,D]
    V = torch.cat([v_cache[:, layer, :cur - 1].float(), v[:, None]], 1)
    s = torch.einsum("bhd,bthd->bht", q, K) / math.sqrt(D)
    return torch.einsum("bht,bthd->bhd", s.softmax(-1), V)


def main():
    B, H, D, M, L, layer, cached = 1, 32, 128, 1024, 4, 0, 511
    dev = torch.device("cuda")
    torch.manual_seed(0)
    qkv = torch.randn(B, 3, H, D, dtype=torch.float16, device=dev)
    zeros = torch.zeros(H, D, dtype=torch.float16, device=dev)
    kc = torch.randn(B, L, M, H, D, dtype=torch.float16, device=dev)
    vc = torch.randn(B, L, M, H, D, dtype=torch.float16, device=dev)
    seq_len = torch.full((B,), cached, dtype=torch.int32, device=dev)
    o = torch.zeros(B, H, D, dtype=torch.float16, device=dev)
    want = torch_decode(qkv, kc, vc, cached + 1, layer)
    o = star_flash_attn.mha_fwd_cuda(qkv, zeros, zeros, zeros, kc, vc, seq_len, o,
                                     B, M, H, D, D, M, L, layer)
    star_flash_attn.check_errors()
    err = (o.float() - want).abs().max().item()
    print("first head:", o[0, 0, :8].tolist())
    print("last head :", o[0, -1, :8].tolist())
    print(f"max |star_flash_attn - torch| = {err:.3e}")
    assert err < 2e-3
    assert torch.equal(vc[:, layer, cached], qkv[:, 2])          # v appended in place


if __name__ == "__main__":
    main()
