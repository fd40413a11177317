"""Every BASELINE.json config at its FULL size on the GPU.

The CPU oracle cannot evaluate a whole full-size problem in seconds, so each config is checked through
(1) slices of (batch, head) pairs against the fp64 oracle -- chosen at the first, the last and a middle
index, so the high ones sit behind the 2^31 / 2^32-element marks of the decode caches -- and (2) properties
that hold at any size: all-equal V rows reproduce that row, a causal prefix is bit-identical to the
truncated problem, all-ones decode gives all ones, the appended cache row is exact and nothing else of
that sequence's cache changes.  configs[1] (B=8 H=16 S=1024 D=64) is small enough for the whole output.

Tolerances as everywhere: bf16 atol = rtol = 1.6e-2 against the fp64 oracle on the same
bf16-rounded inputs.
"""
import numpy as np
import pytest
import torch

from oracle import decode_ref, sdpa_ref

pytestmark = pytest.mark.gpu
TOL = 1.6e-2
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    import starflashattention_amd as m
    m._lib.load()
    return m


def randn_bf16(shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(shape, generator=g, device=DEV, dtype=torch.float32).bfloat16()


def f32(t):
    return t.float().cpu().numpy()


def test_config1_whole_output(sfa):
    """configs[1]: batch=8 heads=16 seqlen=1024 hdim=64 bf16 fwd, non-causal -- the whole output."""
    B, H, S, D = 8, 16, 1024, 64
    q, k, v = (randn_bf16((B, H, S, D), 100 + i) for i in range(3))
    o = sfa.flash_attn_fwd(q, k, v, causal=False)
    torch.cuda.synchronize()
    want = sdpa_ref(f32(q), f32(k), f32(v), causal=False)
    np.testing.assert_allclose(f32(o), want, atol=TOL, rtol=TOL)


def _prefill_full_size(sfa, B, H, S, D, causal, slices, seed):
    q, k, v = (randn_bf16((B, H, S, D), seed + i) for i in range(3))
    # (1) constant V: every output row equals that row (the softmax weights sum to one)
    vrow = randn_bf16((1, 1, 1, D), seed + 7)
    o = sfa.flash_attn_fwd(q, k, vrow.expand(B, H, S, D).contiguous(), causal=causal)
    torch.cuda.synchronize()
    err = (o.float() - vrow.float()).abs().amax().item()
    assert err <= TOL * (1 + vrow.float().abs().amax().item()), err
    del o
    # (2) slices against the fp64 oracle
    full = sfa.flash_attn_fwd(q, k, v, causal=causal)
    torch.cuda.synchronize()
    assert torch.isfinite(full.float()).all()
    for b, h in slices:
        want = sdpa_ref(f32(q[b:b + 1, h:h + 1]), f32(k[b:b + 1, h:h + 1]), f32(v[b:b + 1, h:h + 1]), causal=causal)
        np.testing.assert_allclose(f32(full[b:b + 1, h:h + 1]), want, atol=TOL, rtol=TOL, err_msg=f"(b,h)=({b},{h})")
    return q, k, v, full


def test_config2_headline_shape(sfa):
    """configs[2]: batch=16 heads=32 seqlen=4096 hdim=128 bf16 causal -- the shape bench.py times."""
    B, H, S, D = 16, 32, 4096, 128
    q, k, v, full = _prefill_full_size(sfa, B, H, S, D, True, [(0, 0), (15, 31), (7, 13)], 200)
    # (3) causal prefix: row i depends on keys <= i only, so the first 1024 rows of every head are
    # bit-identical to the 1024-token problem when the SAME kernel serves it (same tile order per row).  The
    # library's own choice for the short problem is the 8-wave kernel: equal to a rounding flip.
    qs, ks, vs = (t[:, :, :1024].contiguous() for t in (q, k, v))
    part_auto = sfa.flash_attn_fwd(qs, ks, vs, causal=True)
    sfa.debug_set("prefill_impl", 40)           # the 4-wave kernel, as the full-size launch above
    try:
        part = sfa.flash_attn_fwd(qs, ks, vs, causal=True)
        torch.cuda.synchronize()
    finally:
        sfa.debug_set("prefill_impl", -1)
    assert torch.equal(full[:, :, :1024], part)
    np.testing.assert_allclose(f32(part_auto), f32(part), atol=TOL / 2, rtol=TOL / 2)


def test_config4_per_gpu_shard(sfa):
    """configs[4]: batch=128 heads=32 seqlen=8192 hdim=128 over 8 GPUs = batch 16 per GPU, non-causal."""
    _prefill_full_size(sfa, 16, 32, 8192, 128, False, [(0, 0), (15, 31)], 300)


def test_config3_decode_full_size(sfa):
    """configs[3]: decode batch=256 seqlen_k=8192 heads=32 hdim=128 bf16 on 2 x 16 GiB caches.  K alone has
    8.6e9 elements: every (b >= 64) slice lies beyond 2^31 elements, (b >= 128) beyond 2^32."""
    B, H, M, D, L = 256, 32, 8192, 128, 1
    pos = M - 1
    dt = torch.bfloat16
    z = torch.zeros(0, dtype=dt, device=DEV)
    sl = torch.full((B,), pos, dtype=torch.int32, device=DEV)
    kc = torch.empty((B, L, M, H, D), dtype=dt, device=DEV)
    vc = torch.empty_like(kc)
    # (1) the reference's known answer at full size: all ones in -> all ones out
    qkv = torch.ones((B, 3, H, D), dtype=dt, device=DEV)
    kc.fill_(1.0)
    vc.fill_(1.0)
    o = torch.zeros((B, H, D), dtype=dt, device=DEV)
    sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, L, 0)
    sfa.check_decode_status()
    assert (o.float() - 1.0).abs().amax().item() <= 1e-2
    # (2) random caches, filled on the device batch by batch
    g = torch.Generator(device=DEV).manual_seed(400)
    for t in (kc, vc):
        for b in range(B):
            t[b].copy_(torch.randn(t[b].shape, generator=g, device=DEV, dtype=torch.float32))
    qkv = randn_bf16((B, 3, H, D), 401)
    watch = [0, 128, 255]
    before = {b: (kc[b].clone(), vc[b].clone()) for b in watch}
    sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, L, 0)
    sfa.check_decode_status()
    assert torch.isfinite(o.float()).all()
    for b, h in [(0, 0), (255, 31), (128, 17)]:
        kb, vb = before[b]
        ref = decode_ref(f32(qkv[b:b + 1, :, h:h + 1]), f32(kb[None, :, :, h:h + 1]), f32(vb[None, :, :, h:h + 1]),
                         [pos], 0, D, dtype="bf16")
        np.testing.assert_allclose(f32(o[b, h]), ref["o"][0, 0], atol=TOL, rtol=TOL, err_msg=f"(b,h)=({b},{h})")
        # the appended rows: V exact, K within one storage ulp (on-device sincosf vs numpy)
        np.testing.assert_array_equal(f32(vc[b, 0, pos, h]), ref["v_row"][0, 0])
        krow = f32(kc[b, 0, pos, h])
        assert np.all(np.abs(krow - ref["k_row"][0, 0]) <= 2.0 ** -7 * np.maximum(1.0, np.abs(krow)) * 1.01)
    for b in watch:         # nothing else of that sequence's cache moved
        kb, vb = before[b]
        assert torch.equal(kc[b, :, :pos], kb[:, :pos]) and torch.equal(vc[b, :, :pos], vb[:, :pos])
        assert torch.equal(vc[b, 0, pos], qkv[b, 2])                        # the whole appended V row, every head
