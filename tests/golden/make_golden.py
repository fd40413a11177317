#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the AUTHORING container only).

Decode vectors come from the reference's own pure-PyTorch ground truth,
/root/reference/examples/python/testFlashDecoder.py (LlamaAttention py:61-94,
apply_rotary_pos_emb py:46-58, LlamaRotaryEmbedding py:7-26), imported from
where it lies with the missing CUDA extension stubbed in sys.modules; nothing
from the reference is copied into this repo -- only inputs and expected outputs
(data) are saved.  The reference cannot travel to the GPU box, so the tests
read these .npz files instead.

Prefill vectors: the reference has no prefill path; expected outputs are
PyTorch eager F.scaled_dot_product_attention in fp32 on the CPU (the
north_star's CPU baseline), on bf16-representable inputs.

Usage:  python tests/golden/make_golden.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PY = "/root/reference/examples/python/testFlashDecoder.py"


def load_reference():
    sys.modules.setdefault("star_flash_attn", types.ModuleType("star_flash_attn"))
    spec = importlib.util.spec_from_file_location("ref_flash_decoder", REF_PY)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # main() is guarded; nothing runs
    return mod


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):   # the reference prints a lot
        return fn(*a, **kw)


def representable(shape, gen):
    """N(0,1) values exactly representable in BOTH fp16 and bf16."""
    x = torch.randn(shape, generator=gen, dtype=torch.float32)
    x = x.to(torch.bfloat16).to(torch.float32)
    x = torch.where(x.abs() < 2.0 ** -14, torch.zeros_like(x), x)   # fp16 normals only
    assert torch.equal(x, x.to(torch.float16).to(torch.float32))
    return x


def bits_bf16(x):
    return (x.contiguous().view(torch.int32).numpy().view(np.uint32) >> 16).astype(np.uint16)


def make_decode(ref):
    gen = torch.Generator().manual_seed(20260410)
    B, H, D, L, M = 2, 2, 128, 2, 136
    layer = 1
    seq_lens = [0, 1, 31, 32, 33, 127, 128, 129, 135]     # tokens already cached
    qkv = representable((B, 3, H, D), gen)
    kc = representable((B, L, M, H, D), gen)
    vc = representable((B, L, M, H, D), gen)
    out = dict(qkv_bf16bits=bits_bf16(qkv), k_cache_bf16bits=bits_bf16(kc),
               v_cache_bf16bits=bits_bf16(vc), seq_lens=np.array(seq_lens, np.int32),
               idx_layer=np.int32(layer), dims=np.array([B, H, D, L, M], np.int32))
    attn = ref.LlamaAttention(D)
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        os_, ks_, qs_ = [], [], []
        for s in seq_lens:
            cur = s + 1          # oracle counts the new token
            a, b, c = qkv.to(dt), kc.to(dt), vc.to(dt)
            o = quiet(attn.forward, a, b, c, cur, layer)            # [B,H,1,D]
            cos, sin = quiet(attn.rotary_emb, cur, a.device)
            q_, k_, _ = torch.chunk(a, 3, dim=1)
            qr, kr = ref.apply_rotary_pos_emb(q_, k_, cos[-1:], sin[-1:])
            os_.append(o[:, :, 0, :].float().numpy())
            ks_.append(kr[:, 0].float().numpy())                     # [B,H,D]
            qs_.append(qr[:, 0].float().numpy())
        out[f"o_{tag}"] = np.stack(os_)          # [case,B,H,D]
        out[f"k_rot_{tag}"] = np.stack(ks_)
        out[f"q_rot_{tag}"] = np.stack(qs_)
    # partial rotary (rot_dim < D): the reference oracle always rotates the whole
    # head, so pin only the rotated rows, built from the reference's own
    # LlamaRotaryEmbedding(dim=rot_dim) + apply_rotary_pos_emb on the leading slice
    # (kernel-side rule: flash_attn.cu:193,209).
    rot = 64
    remb = ref.LlamaRotaryEmbedding(dim=rot)
    pk, pq = [], []
    for s in seq_lens:
        cos, sin = quiet(remb, s + 1, qkv.device)
        q_, k_, _ = torch.chunk(qkv, 3, dim=1)
        qr, kr = ref.apply_rotary_pos_emb(q_[..., :rot], k_[..., :rot], cos[-1:], sin[-1:])
        pq.append(torch.cat([qr, q_[..., rot:]], -1)[:, 0].numpy())
        pk.append(torch.cat([kr, k_[..., rot:]], -1)[:, 0].numpy())
    out["partial_rot_dim"] = np.int32(rot)
    out["partial_q_rot_f32"] = np.stack(pq)
    out["partial_k_rot_f32"] = np.stack(pk)
    np.savez_compressed(os.path.join(HERE, "decode_llama_ref.npz"), **out)
    print("decode_llama_ref.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def make_ones_kat():
    """The reference's only known answer: all-ones in -> all 1.0 out
    (examples/cpp/testFlashDecoder.cc:63-78,116-129; shapes cc:138-146).  Recorded
    as metadata only -- the tensors are trivially regenerated."""
    cases = np.array([[512, 511], [1024, 1023], [2048, 2047], [8192, 6143], [8192, 8191]], np.int32)
    bad = np.array([[4096, 4096]], np.int32)     # cc:141-142 pair 4 overruns the cache
    np.savez_compressed(os.path.join(HERE, "decode_ones_kat.npz"),
                        max_seq_len__seq_len=cases, must_raise=bad,
                        dims_BHDL=np.array([2, 32, 128, 4], np.int32), expect=np.float32(1.0))
    print("decode_ones_kat.npz")


def make_prefill():
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(4096128)
    out = {}
    cases = [  # name, B, Hq, Hkv, Sq, Sk, D
        ("s129_d128", 1, 1, 1, 129, 129, 128),
        ("s200_d64", 1, 1, 1, 200, 200, 64),
        ("gqa_s64_d128", 1, 2, 1, 64, 64, 128),
    ]
    names = []
    for name, B, Hq, Hkv, Sq, Sk, D in cases:
        q = representable((B, Hq, Sq, D), gen)
        k = representable((B, Hkv, Sk, D), gen)
        v = representable((B, Hkv, Sk, D), gen)
        g = Hq // Hkv
        kk, vv = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        out[f"{name}_q"] = bits_bf16(q)
        out[f"{name}_k"] = bits_bf16(k)
        out[f"{name}_v"] = bits_bf16(v)
        for causal in (False, True):
            o = F.scaled_dot_product_attention(q, kk, vv, is_causal=causal)
            out[f"{name}_o_{'causal' if causal else 'full'}"] = o.numpy()
        names.append(name)
    out["names"] = np.array(names)
    # BASELINE.json configs[0]: the reference-side CPU-runnable plumbing case.
    g2 = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((1, 4, 128, 64), generator=g2) for _ in range(3))
    # inputs are regenerated from torch.manual_seed(0) in the test; pin them by checksum
    out["cfg0_input_sums"] = np.array([q.double().sum().item(), k.double().sum().item(),
                                       v.double().sum().item()])
    out["cfg0_o_full"] = F.scaled_dot_product_attention(q, k, v).numpy()
    np.savez_compressed(os.path.join(HERE, "prefill_sdpa_cpu.npz"), **out)
    print("prefill_sdpa_cpu.npz", names)


if __name__ == "__main__":
    torch.set_num_threads(4)
    ref = load_reference()
    make_decode(ref)
    make_ones_kat()
    make_prefill()
