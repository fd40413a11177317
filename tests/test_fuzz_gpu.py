"""Randomised shape sweep of both hot paths through the C ABI against the fp64 oracle (seeded: the same
cases every run).  Complements the hand-picked edge cases in test_prefill_gpu.py / test_decode_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import decode_ref, round_to, sdpa_ref

pytestmark = pytest.mark.gpu
TOL = {"fp16": 2e-3, "bf16": 1.6e-2}
TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def sfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    import starflashattention_amd as m
    m._lib.load()
    return m


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_prefill(sfa, seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = ("bf16", "fp16")[seed % 2]
    D = (128, 64, 256)[int(rng.integers(3))]
    Hkv = int(rng.integers(1, 4))
    Hq = Hkv * int(rng.choice([1, 1, 2, 4]))
    B = int(rng.integers(1, 4))
    Sq = int(rng.integers(1, 700))
    Sk = Sq if rng.random() < 0.5 else int(rng.integers(1, 900))
    causal = bool(rng.integers(2))
    fast = bool(rng.integers(2))
    scale = float(rng.choice([0.0, 0.05, 0.2]))            # 0 -> default 1/sqrt(D)
    q = round_to(rng.standard_normal((B, Hq, Sq, D)), dtype)
    k = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    v = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    want = sdpa_ref(q, k, v, causal=causal, scale=scale if scale else None)
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(x).to(TDT[dtype]).to(dev)
    # a [B, S, H, D]-stored tensor viewed [B, H, S, D] half of the time
    def lay(x):
        x = t(x)
        return x.transpose(1, 2).contiguous().transpose(1, 2) if rng.random() < 0.5 else x
    # head_dim 128: half of the cases on the 4-wave persistent kernel (the library's own choice for large problems only)
    impl = 40 if D == 128 and rng.random() < 0.5 else -1
    sfa.debug_set("prefill_impl", impl)
    try:
        o = sfa.flash_attn_fwd(lay(q), lay(k), lay(v), causal=causal, softmax_scale=scale if scale else None,
                               fast_scale=fast)
        torch.cuda.synchronize()
    finally:
        sfa.debug_set("prefill_impl", -1)
    np.testing.assert_allclose(o.float().cpu().numpy(), want, atol=TOL[dtype], rtol=TOL[dtype],
                               err_msg=f"B={B} Hq={Hq} Hkv={Hkv} Sq={Sq} Sk={Sk} D={D} causal={causal} fast={fast} impl={impl}")


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_decode(sfa, seed):
    rng = np.random.default_rng(2000 + seed)
    dtype = ("fp16", "bf16")[seed % 2]
    D = int(rng.choice([64, 128, 128, 256]))
    H = int(rng.integers(1, 7))
    B = int(rng.integers(1, 5))
    L = int(rng.integers(1, 3))
    layer = int(rng.integers(L))
    M = 16 * int(rng.integers(2, 40))
    rot = int(rng.choice([0, D // 2, D]))
    splits = int(rng.choice([0, 1, 2, 5]))
    layout = str(rng.choice(["blmhd", "blhmd", "paged"]))
    lens = [int(x) for x in rng.integers(0, M, B)]
    qkv = round_to(rng.standard_normal((B, 3, H, D)), dtype)
    kc = round_to(rng.standard_normal((B, L, M, H, D)), dtype)
    vc = round_to(rng.standard_normal((B, L, M, H, D)), dtype)
    ref = decode_ref(qkv, kc.copy(), vc.copy(), lens, layer, rot, dtype=dtype)
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(TDT[dtype]).to(dev)
    kw = {"num_splits": splits}
    k_t, v_t = t(kc), t(vc)
    if layout == "blhmd":
        k_t, v_t = (x.permute(0, 1, 3, 2, 4).contiguous() for x in (k_t, v_t))
        kw["kv_layout"] = "blhmd"
    elif layout == "paged":
        ps = 16
        pps = M // ps
        table = torch.from_numpy(rng.permutation(B * pps).astype(np.int32)).view(B, pps).to(dev)
        def to_pool(c):
            pool = torch.zeros((B * pps, L, ps, H, D), dtype=TDT[dtype], device=dev)
            pool[table.long().view(-1)] = c.view(B, L, pps, ps, H, D).permute(0, 2, 1, 3, 4, 5).reshape(B * pps, L, ps, H, D)
            return pool
        k_t, v_t = to_pool(k_t), to_pool(v_t)
        kw.update(kv_layout="paged", block_table=table)
    o = torch.empty((B, H, D), dtype=TDT[dtype], device=dev)
    z = torch.zeros(0, dtype=TDT[dtype], device=dev)
    sfa.flash_decode(t(qkv), z, z, z, k_t, v_t, torch.tensor(lens, dtype=torch.int32, device=dev), o, B, M, H, D, rot,
                     M, L, layer, **kw)
    sfa.check_decode_status()
    np.testing.assert_allclose(o.float().cpu().numpy(), ref["o"], atol=TOL[dtype], rtol=TOL[dtype],
                               err_msg=f"B={B} H={H} D={D} L={L} M={M} rot={rot} splits={splits} {layout} lens={lens}")


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_decode_grouped(sfa, seed):
    """Grouped queries (VALU and matrix-core kernels, all layouts) against the oracle on the expanded problem."""
    rng = np.random.default_rng(3000 + seed)
    dtype = ("fp16", "bf16")[seed % 2]
    D = int(rng.choice([64, 128, 128, 128, 256]))
    G = int(rng.choice([2, 4, 8, 16]))
    force = int(rng.choice([-1, -1, 0, 1]))     # -1: the library's choice of kernel; 0 / 1: VALU / matrix-core where both exist
    Hkv = int(rng.integers(1, 4))
    H = Hkv * G
    B = int(rng.integers(1, 4))
    M = 16 * int(rng.integers(2, 30))
    rot = int(rng.choice([0, D]))
    splits = int(rng.choice([0, 1, 3]))
    layout = str(rng.choice(["blmhd", "blhmd", "paged"]))
    lens = [int(x) for x in rng.integers(0, M, B)]
    qkv = round_to(rng.standard_normal((B, H + 2 * Hkv, D)), dtype)
    kc = round_to(rng.standard_normal((B, 1, M, Hkv, D)), dtype)
    vc = round_to(rng.standard_normal((B, 1, M, Hkv, D)), dtype)
    rep = lambda x, ax: np.repeat(x, G, axis=ax)
    qkv_x = np.stack([qkv[:, :H], rep(qkv[:, H:H + Hkv], 1), rep(qkv[:, H + Hkv:], 1)], 1)
    ref = decode_ref(qkv_x, rep(kc, 3), rep(vc, 3), lens, 0, rot, dtype=dtype)
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(TDT[dtype]).to(dev)
    kw = {"num_splits": splits, "num_heads_kv": Hkv}
    k_t, v_t = t(kc), t(vc)
    if layout == "blhmd":
        k_t, v_t = (x.permute(0, 1, 3, 2, 4).contiguous() for x in (k_t, v_t))
        kw["kv_layout"] = "blhmd"
    elif layout == "paged":
        ps = 16
        pps = M // ps
        table = torch.from_numpy(rng.permutation(B * pps).astype(np.int32)).view(B, pps).to(dev)
        def to_pool(c):
            pool = torch.zeros((B * pps, 1, ps, Hkv, D), dtype=TDT[dtype], device=dev)
            pool[table.long().view(-1)] = c.view(B, 1, pps, ps, Hkv, D).permute(0, 2, 1, 3, 4, 5).reshape(B * pps, 1, ps, Hkv, D)
            return pool
        k_t, v_t = to_pool(k_t), to_pool(v_t)
        kw.update(kv_layout="paged", block_table=table)
    o = torch.empty((B, H, D), dtype=TDT[dtype], device=dev)
    z = torch.zeros(0, dtype=TDT[dtype], device=dev)
    sfa.debug_set("decode_gqa_mfma", force)
    try:
        sfa.flash_decode(t(qkv), z, z, z, k_t, v_t, torch.tensor(lens, dtype=torch.int32, device=dev), o, B, M, H, D, rot,
                         M, 1, 0, **kw)
        sfa.check_decode_status()
    finally:
        sfa.debug_set("decode_gqa_mfma", -1)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref["o"], atol=TOL[dtype], rtol=TOL[dtype],
                               err_msg=f"B={B} H={H} Hkv={Hkv} D={D} M={M} rot={rot} splits={splits} {layout} lens={lens} force={force}")


def test_fuzz_w4_chained_qtiles_against_the_128_row_kernel():
    """tools/w4_fuzz.py: 96 seeded shapes with more q-tiles than persistent workgroups (every seam path of the 4-wave kernel)
    against the 128-row kernel, output and log-sum-exp."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "w4_fuzz.py"), "96"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 mismatch(es) in 96 cases" in r.stdout
