"""Pin the oracle (oracle/) against the reference-generated golden vectors (CPU only).

decode_llama_ref.npz was produced by the reference's own LlamaAttention
(tests/golden/make_golden.py); the only known answer the reference's tests hold
is "all ones -> all 1.0" (examples/cpp/testFlashDecoder.cc:63-78,116-129).
"""
import numpy as np
import pytest

from conftest import bf16bits_to_f32
from oracle import decode_ref, sdpa_ref, rotary_table_ref, bf16_round, fp16_round


def _decode_inputs(g):
    return (bf16bits_to_f32(g["qkv_bf16bits"]), bf16bits_to_f32(g["k_cache_bf16bits"]),
            bf16bits_to_f32(g["v_cache_bf16bits"]), int(g["idx_layer"]))


@pytest.mark.parametrize("case", range(9))
def test_decode_oracle_matches_reference_fp32(decode_golden, case):
    g = decode_golden
    qkv, kc, vc, layer = _decode_inputs(g)
    s = int(g["seq_lens"][case])
    B, H, D, L, M = g["dims"]
    r = decode_ref(qkv, kc.copy(), vc.copy(), [s] * B, layer, D, dtype="fp32",
                   round_q=False, round_kv=False)
    # fp64 restatement vs the reference evaluated in fp32
    np.testing.assert_allclose(r["o"], g["o_f32"][case], atol=1e-5, rtol=1e-5)
    # rotated rows: the fp32 angle pos*inv_freq carries ~1 ulp(angle) of libm noise
    # (numpy pow vs torch pow), i.e. up to ~1e-5 * |x| at pos ~ 135
    np.testing.assert_allclose(r["k_row"], g["k_rot_f32"][case], atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(r["q_rot"], g["q_rot_f32"][case], atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("dtype,tag,ulp,otol", [("fp16", "f16", 2.0 ** -10, 3e-3),
                                                ("bf16", "bf16", 2.0 ** -7, 2e-2)])
def test_decode_oracle_matches_reference_16bit(decode_golden, dtype, tag, ulp, otol):
    """Reference run natively in fp16/bf16 on the CPU: the appended K row must agree to one
    storage ulp (cos/sin come from different libms); the output to the 16-bit softmax noise."""
    g = decode_golden
    qkv, kc, vc, layer = _decode_inputs(g)
    B, H, D, L, M = g["dims"]
    for case, s in enumerate(g["seq_lens"]):
        r = decode_ref(qkv, kc.copy(), vc.copy(), [int(s)] * B, layer, D, dtype=dtype)
        gk = g[f"k_rot_{tag}"][case]
        assert np.all(np.abs(r["k_row"] - gk) <= ulp * np.maximum(1.0, np.abs(gk)) * 1.01)
        assert np.mean(r["k_row"] == gk) > 0.99
        np.testing.assert_allclose(r["o"], g[f"o_{tag}"][case], atol=otol, rtol=otol)


def test_decode_oracle_ragged_and_append(decode_golden):
    g = decode_golden
    qkv, kc, vc, layer = _decode_inputs(g)
    B, H, D, L, M = g["dims"]
    lens = [int(g["seq_lens"][2]), int(g["seq_lens"][6])]      # 31 and 128
    kc2, vc2 = kc.copy(), vc.copy()
    r = decode_ref(qkv, kc2, vc2, lens, layer, D, dtype="fp32", round_q=False, round_kv=False)
    np.testing.assert_allclose(r["o"][0], g["o_f32"][2][0], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(r["o"][1], g["o_f32"][6][1], atol=2e-6, rtol=1e-5)
    # caches mutated only at [b, layer, seq_len[b]]
    diff = (kc2 != kc) | (vc2 != vc)
    where = np.argwhere(diff.any(axis=(3, 4)))
    assert {tuple(w) for w in where} == {(0, layer, lens[0]), (1, layer, lens[1])}
    np.testing.assert_array_equal(vc2[0, layer, lens[0]], qkv[0, 2])


def test_decode_oracle_partial_rotary(decode_golden):
    g = decode_golden
    qkv, kc, vc, layer = _decode_inputs(g)
    B, H, D, L, M = g["dims"]
    rot = int(g["partial_rot_dim"])
    for case, s in enumerate(g["seq_lens"]):
        r = decode_ref(qkv, kc.copy(), vc.copy(), [int(s)] * B, layer, rot, dtype="fp32",
                       round_q=False, round_kv=False)
        np.testing.assert_allclose(r["k_row"], g["partial_k_rot_f32"][case], atol=5e-5, rtol=1e-5)
        np.testing.assert_allclose(r["q_rot"], g["partial_q_rot_f32"][case], atol=5e-5, rtol=1e-5)
        np.testing.assert_array_equal(r["k_row"][..., rot:], qkv[:, 1, :, rot:])


def test_decode_oracle_ones_known_answer(ones_kat):
    """cc:63-78: all-ones qkv and caches -> every output element is 1.0 (any seq_len);
    RoPE on q,k does not matter because V is constant."""
    B, H, D, L = (2, 4, 128, 2)     # the KAT holds for any B,H; keep the CPU case small
    for M, s in ones_kat["max_seq_len__seq_len"][:3]:
        qkv = np.ones((B, 3, H, D), np.float32)
        kc = np.ones((B, L, int(M), H, D), np.float32)
        vc = np.ones_like(kc)
        r = decode_ref(qkv, kc, vc, [int(s)] * B, 0, D, dtype="fp16")
        np.testing.assert_allclose(r["o"], float(ones_kat["expect"]), atol=1e-6)
    M, s = ones_kat["must_raise"][0]
    with pytest.raises(ValueError):
        decode_ref(np.ones((1, 3, 1, 8), np.float32), np.ones((1, 1, 4, 1, 8), np.float32),
                   np.ones((1, 1, 4, 1, 8), np.float32), [4], 0, 8)


def test_decode_oracle_bias_and_table():
    rng = np.random.default_rng(3)
    B, H, D, L, M = 1, 2, 64, 1, 20
    qkv = fp16_round(rng.standard_normal((B, 3, H, D)))
    kc = fp16_round(rng.standard_normal((B, L, M, H, D)))
    vc = fp16_round(rng.standard_normal((B, L, M, H, D)))
    bias = fp16_round(rng.standard_normal((3, H, D)))
    # bias == shifting qkv beforehand (in exact arithmetic)
    r1 = decode_ref(qkv, kc.copy(), vc.copy(), [7], 0, D, dtype="fp32", q_bias=bias[0],
                    k_bias=bias[1], v_bias=bias[2], round_q=False, round_kv=False)
    r2 = decode_ref(qkv + bias[None], kc.copy(), vc.copy(), [7], 0, D, dtype="fp32",
                    round_q=False, round_kv=False)
    np.testing.assert_allclose(r1["o"], r2["o"], atol=1e-6)
    # an fp32 LUT reproduces the on-the-fly trig
    c, s = rotary_table_ref(M, D, dtype="fp32")
    r3 = decode_ref(qkv, kc.copy(), vc.copy(), [7], 0, D, dtype="fp32", cos_table=c, sin_table=s,
                    round_q=False, round_kv=False)
    r4 = decode_ref(qkv, kc.copy(), vc.copy(), [7], 0, D, dtype="fp32", round_q=False, round_kv=False)
    np.testing.assert_allclose(r3["o"], r4["o"], atol=1e-6)
    assert c.shape == (M, D // 2) and np.all(c[0] == 1.0) and np.all(s[0] == 0.0)


@pytest.mark.parametrize("name", ["s129_d128", "s200_d64", "gqa_s64_d128"])
@pytest.mark.parametrize("causal", [False, True])
def test_sdpa_oracle_matches_torch_cpu(prefill_golden, name, causal):
    g = prefill_golden
    q, k, v = (bf16bits_to_f32(g[f"{name}_{t}"]) for t in "qkv")
    o = sdpa_ref(q, k, v, causal=causal)
    np.testing.assert_allclose(o, g[f"{name}_o_{'causal' if causal else 'full'}"], atol=3e-6, rtol=1e-5)


def test_sdpa_oracle_config0_plumbing(prefill_golden):
    """BASELINE.json configs[0]: B=1 H=4 S=128 D=64 fp32, PyTorch eager SDPA on the CPU."""
    import torch
    from oracle import sdpa_torch_cpu
    g2 = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((1, 4, 128, 64), generator=g2) for _ in range(3))
    sums = np.array([t.double().sum().item() for t in (q, k, v)])
    np.testing.assert_allclose(sums, prefill_golden["cfg0_input_sums"], rtol=1e-12)
    o_t = sdpa_torch_cpu(q, k, v).numpy()
    np.testing.assert_allclose(o_t, prefill_golden["cfg0_o_full"], atol=1e-6)
    np.testing.assert_allclose(sdpa_ref(q.numpy(), k.numpy(), v.numpy()), o_t, atol=3e-6, rtol=1e-5)


def test_sdpa_oracle_rectangular_causal_and_lse():
    rng = np.random.default_rng(5)
    q = rng.standard_normal((1, 2, 5, 16)); k = rng.standard_normal((1, 2, 9, 16)); v = rng.standard_normal((1, 2, 9, 16))
    o, lse = sdpa_ref(q, k, v, causal=True, return_lse=True)
    # bottom-right alignment: the last query sees every key
    full = sdpa_ref(q[:, :, -1:], k, v)
    np.testing.assert_allclose(o[:, :, -1:], full, atol=1e-6)
    # first query sees keys 0..4
    part = sdpa_ref(q[:, :, :1], k[:, :, :5], v[:, :, :5])
    np.testing.assert_allclose(o[:, :, :1], part, atol=1e-6)
    assert np.all(np.isfinite(lse))


def test_rounding_helpers():
    x = np.array([1.0, 1.00390625, 1.0 + 2.0 ** -8, 3.14159, -2.71828, 65504.0, 1e-8], np.float32)
    import torch
    t = torch.from_numpy(x)
    np.testing.assert_array_equal(bf16_round(x), t.to(torch.bfloat16).float().numpy())
    np.testing.assert_array_equal(fp16_round(x), t.to(torch.float16).float().numpy())
