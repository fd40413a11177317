"""GPU parity of the HIP prefill forward (through the C ABI) against the CPU oracle and the
golden vectors (PyTorch eager SDPA on the CPU, tests/golden/make_golden.py).

Tolerance: kernel vs fp64 oracle on identically rounded inputs --
fp16 atol=rtol=2e-3, bf16 atol=rtol=1.6e-2; additionally the kernel's error must stay within 2x
that of a plain same-dtype PyTorch computation.
"""
import numpy as np
import pytest
import torch

from conftest import bf16bits_to_f32
from oracle import sdpa_ref, round_to

pytestmark = pytest.mark.gpu

TOL = {"fp16": 2e-3, "bf16": 1.6e-2}
TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def sfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    import starflashattention_amd as m
    m._lib.load()
    return m


def run_fwd(sfa, q, k, v, dtype, causal, **kw):
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(TDT[dtype]).to(dev)
    out = sfa.flash_attn_fwd(t(q), t(k), t(v), causal=causal, **kw)
    torch.cuda.synchronize()
    if isinstance(out, tuple):
        return out[0].float().cpu().numpy(), out[1].cpu().numpy()
    return out.float().cpu().numpy()


@pytest.mark.parametrize("name", ["s129_d128", "s200_d64", "gqa_s64_d128"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_prefill_golden(sfa, prefill_golden, name, causal, dtype):
    g = prefill_golden
    q, k, v = (bf16bits_to_f32(g[f"{name}_{t}"]) for t in "qkv")     # representable in both dtypes
    o = run_fwd(sfa, q, k, v, dtype, causal)
    want = g[f"{name}_o_{'causal' if causal else 'full'}"]
    np.testing.assert_allclose(o, want, atol=TOL[dtype], rtol=TOL[dtype])


CASES = [  # B, Hq, Hkv, Sq, Sk, D
    (1, 1, 1, 1, 1, 128),          # single token
    (2, 3, 3, 64, 64, 128),
    (1, 2, 2, 256, 256, 128),      # exactly one q-tile, 4 kv tiles
    (1, 2, 1, 257, 257, 128),      # one row into the second q-tile, GQA
    (2, 2, 2, 513, 513, 64),
    (1, 4, 2, 300, 300, 64),
    (1, 2, 2, 100, 333, 128),      # Sq < Sk: bottom-right aligned causal
    (1, 1, 1, 333, 100, 128),      # Sq > Sk: the first rows see no key under the causal mask
    (1, 8, 8, 1024, 1024, 128),
    (1, 2, 2, 1280, 1280, 128),    # 5 q-tiles: two pairs + an unpaired middle tile
    (1, 1, 1, 1800, 1800, 64),     # 8 q-tiles, ragged last tile
    (2, 4, 2, 1100, 1100, 128),    # several heads per XCD list of the persistent kernel, ragged, GQA
    (1, 10, 10, 700, 700, 128),    # more heads than XCDs: persistent workgroups walk several units
    (1, 2, 1, 300, 300, 256),      # head_dim 256 (its own kernel, whatever prefill_impl says), GQA, ragged
    (2, 3, 3, 129, 400, 256),      # Sq < Sk
    # more q-tiles than persistent workgroups (36 units per XCD list against 32 slots) and very few keys: the 4-wave kernel
    # chains q-tiles of one, two, three and five tiles -- the seam half-step, the Q fetch in the gaps and its fallback at the
    # top of a q-tile, the Q request at the top of q-tiles of one or two tiles; causal: most q-tiles see no key at all
    (2, 16, 16, 2304, 64, 128),
    (2, 16, 16, 2304, 128, 128),
    (2, 16, 4, 2304, 192, 128),
    (2, 16, 16, 2304, 320, 128),
]


# every kernel the dispatcher can pick must pass on its own: the 4-wave persistent kernel (head_dim 128
# and 256), the 8-wave 256-row pipelined kernel (one or two q-tile pairs per workgroup) and the 128-row
# geometry for small grids, each in its exact-scale and its prescaled-Q flavour, and the auto choice.
# The earlier generations (baseline, 16x16x32 MFMA) live in the A/B build only: tests/test_variants_gpu.py.
IMPLS = {"auto": -1, "w4": 42, "prescaled_w4": 41, "rows256": 10, "rows256x2": 10, "rows128": 22,
         "prescaled256": 3, "prescaled256x2": 3, "prescaled128": 21, "d256_fallback": 61}
W4_DIMS = (128,)            # head dims the 4-wave kernel serves
OLD_DIMS = (64, 128)            # ... and the 8-wave / 128-row kernels
# prescaled kernels carry Q*scale*log2(e) rounded to 16 bit: the log-sum-exp is good to input
# precision (relative 2^-9 / 2^-12 of the scores), not to the fp32-class 2e-3 of the exact kernels
LSE_TOL = {"exact": {"bf16": 2e-3, "fp16": 2e-3}, "prescaled": {"bf16": 1.5e-2, "fp16": 4e-3}}


@pytest.fixture
def knobs(sfa):
    """Kernel-variant selection through sfa_debug_set (the launch paths read no environment variable);
    everything is back on the library's own choice after the test."""
    names = ("prefill_impl", "prefill_pairs")
    yield sfa.debug_set
    for n in names:
        sfa.debug_set(n, -1)


def select_impl(knobs, impl):
    knobs("prefill_impl", IMPLS[impl])
    knobs("prefill_pairs", 2 if impl.endswith("x2") else 1 if impl.endswith("256") else -1)


def serves(impl, D):
    if impl == "auto":
        return True
    if D == 256:                        # "auto" = the persistent kernel (prefill_w4d_kernel.hip); 61 = the compiler-scheduled
        return impl == "d256_fallback"  # fallback for heads whose rows do not fit 32-bit buffer descriptors
    if impl == "d256_fallback":
        return False
    return D in (W4_DIMS if impl.endswith("w4") else OLD_DIMS)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "b%d_hq%d_hkv%d_sq%d_sk%d_d%d" % c)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("impl", list(IMPLS))
def test_prefill_vs_oracle(sfa, knobs, case, causal, dtype, impl):
    B, Hq, Hkv, Sq, Sk, D = case
    if not serves(impl, D):
        pytest.skip(f"{impl} does not serve head_dim {D}")
    select_impl(knobs, impl)
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    q = round_to(rng.standard_normal((B, Hq, Sq, D)), dtype)
    k = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    v = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    want, lse_want = sdpa_ref(q, k, v, causal=causal, return_lse=True)
    o, lse = run_fwd(sfa, q, k, v, dtype, causal, return_lse=True)
    np.testing.assert_allclose(o, want, atol=TOL[dtype], rtol=TOL[dtype])
    fin = np.isfinite(lse_want)
    ltol = LSE_TOL["prescaled" if impl.startswith("prescaled") else "exact"][dtype]
    np.testing.assert_allclose(lse[fin], lse_want[fin], atol=ltol, rtol=ltol)
    assert np.all(lse[~fin] == -np.inf)
    # error budget: no worse than 2x a plain same-dtype torch computation on the GPU
    dev = torch.device("cuda:0")
    tq, tk, tv = (torch.from_numpy(x).to(TDT[dtype]).to(dev) for x in (q, k, v))
    g = Hq // Hkv
    s = torch.matmul(tq, tk.repeat_interleave(g, 1).transpose(-1, -2)) / (D ** 0.5)
    if causal:
        i = torch.arange(Sq, device=dev)[:, None]
        j = torch.arange(Sk, device=dev)[None, :]
        s = s.masked_fill(~(j <= i + (Sk - Sq)), float("-inf"))
    p = torch.softmax(s.float(), -1).to(TDT[dtype])
    plain = torch.nan_to_num(torch.matmul(p, tv.repeat_interleave(g, 1))).float().cpu().numpy()
    e_kernel = np.abs(o - want).max()
    e_plain = np.abs(plain - want).max()
    assert e_kernel <= 2 * e_plain + 1e-3, (e_kernel, e_plain)


def test_prefill_strided_layouts_and_out(sfa):
    """[B,S,H,D]-stored tensors viewed as [B,H,S,D] (the other common layout) and a caller-owned out."""
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, H, S, D = 2, 3, 200, 128
    q, k, v = (torch.randn(B, S, H, D, device=dev).bfloat16() for _ in range(3))
    out_store = torch.zeros(B, S, H, D, device=dev, dtype=torch.bfloat16)
    o = sfa.flash_attn_fwd(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), causal=True,
                           out=out_store.transpose(1, 2))
    o2 = sfa.flash_attn_fwd(q.transpose(1, 2).contiguous(), k.transpose(1, 2).contiguous(),
                            v.transpose(1, 2).contiguous(), causal=True)
    torch.cuda.synchronize()
    assert o.data_ptr() == out_store.data_ptr()
    assert torch.equal(o, o2)
    # qkv packed [B,S,3,H,D] slices (stride between q,k,v inside one allocation)
    qkv = torch.randn(B, S, 3, H, D, device=dev).half()
    o3 = sfa.flash_attn_fwd(qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2))
    want = sdpa_ref(*(qkv[:, :, i].transpose(1, 2).float().cpu().numpy() for i in range(3)))
    np.testing.assert_allclose(o3.float().cpu().numpy(), want, atol=2e-3, rtol=2e-3)


def test_prefill_auto_falls_back_when_a_head_spans_2gib(sfa):
    """The 4-wave kernel addresses one head's K / V / Q rows through 32-bit buffer descriptors.  A strided view whose
    rows of one head span 2 GiB or more must not fail under the auto rule (round-2 verdict, weak #5): the dispatcher
    sends it to the 8-wave kernel, whose addressing is 64-bit.  Shape otherwise inside the 4-wave kernel's auto range
    (256 q-tiles, full attention); checked against the fp64 oracle on slices."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    B, H, Sq, Sk, D = 1, 1, 256 * 256, 1024, 128
    stride = 1 << 21                                        # 4 MiB between key rows: 1023 rows apart = 4 GiB
    kbase = torch.zeros(Sk * stride, dtype=torch.bfloat16, device=dev)
    k = torch.as_strided(kbase, (B, H, Sk, D), (0, 0, stride, 1))
    k.copy_(torch.randn((B, H, Sk, D), generator=g, device=dev).bfloat16())
    q = torch.randn((B, H, Sq, D), generator=g, device=dev).bfloat16()
    v = torch.randn((B, H, Sk, D), generator=g, device=dev).bfloat16()
    assert (Sk - 1) * stride * 2 >= 2 ** 31
    o = sfa.flash_attn_fwd(q, k, v, causal=False)           # auto choice
    # forcing the 4-wave kernel on this view is refused, which is what the auto rule must not run into
    sfa.debug_set("prefill_impl", 42)
    try:
        with pytest.raises(sfa.SfaError):
            sfa.flash_attn_fwd(q, k, v, causal=False)
    finally:
        sfa.debug_set("prefill_impl", -1)
    torch.cuda.synchronize()
    kf, vf = k.float().cpu().numpy(), v.float().cpu().numpy()
    for r0 in (0, 31 * 256 + 77, Sq - 256):
        want = sdpa_ref(q[:, :, r0:r0 + 256].float().cpu().numpy(), kf, vf, causal=False)
        np.testing.assert_allclose(o[:, :, r0:r0 + 256].float().cpu().numpy(), want, atol=1.6e-2, rtol=1.6e-2)
    del kbase, k


@pytest.mark.parametrize("B,H,Sq,Sk,causal,w4", [
    (2, 32, 4096, 4096, True, True),        # 1024 q-tiles, 16 per head: +6 % (profiles/r03_prefill_crossover.txt)
    (8, 32, 2048, 2048, True, True),        # 2048 q-tiles, 8 per head: +8 %
    (2, 32, 2048, 2048, True, False),       # 512 q-tiles, 8 per head: -2 %
    (16, 32, 1024, 1024, True, True),       # 2048 q-tiles, 4 per head: +2 %
    (8, 32, 4096, 1024, True, False),       # fewer keys than queries under the causal mask: -20 %
    (8, 32, 4096, 8192, True, True),
    (8, 32, 4096, 64, False, True),         # full attention: any key count from 256 q-tiles on
    (1, 8, 4096, 4096, False, False),       # 128 q-tiles
])
def test_prefill_auto_rule_follows_the_measured_crossover(sfa, B, H, Sq, Sk, causal, w4):
    """The dispatcher's choice (read back through sfa_debug_get("last_prefill_kernel")) at the shapes the crossover
    table was measured on, and the chosen kernel's output against a forced other geometry."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.randn((B, H, Sq, 128), generator=g, device=dev).bfloat16()
    k, v = (torch.randn((B, H, Sk, 128), generator=g, device=dev).bfloat16() for _ in range(2))
    o = sfa.flash_attn_fwd(q, k, v, causal=causal)
    assert ("prefill_w4_kernel" in sfa.last_prefill_kernel()) == w4, sfa.last_prefill_kernel()
    sfa.debug_set("prefill_impl", 22 if w4 else 42)
    try:
        o2 = sfa.flash_attn_fwd(q, k, v, causal=causal)
    finally:
        sfa.debug_set("prefill_impl", -1)
    np.testing.assert_allclose(o.float().cpu().numpy(), o2.float().cpu().numpy(), atol=8e-3, rtol=8e-3)


NON_BASELINE = [(i, 128) for i in IMPLS if i not in ("auto", "d256_fallback")] + [("auto", 256), ("d256_fallback", 256)]


@pytest.mark.parametrize("impl,D", NON_BASELINE)
def test_prefill_forced_rescale_branch(sfa, knobs, impl, D):
    """cdna_hip_programming.md rule 26: force the online-softmax max to jump at a chosen tile --
    one K row far larger than the rest, placed late in the sequence, against every Q row."""
    select_impl(knobs, impl)
    rng = np.random.default_rng(9)
    B, H, S = 1, 2, 640
    q = round_to(rng.standard_normal((B, H, S, D)), "bf16")
    k = round_to(0.1 * rng.standard_normal((B, H, S, D)), "bf16")
    v = round_to(rng.standard_normal((B, H, S, D)), "bf16")
    for spike_at in (5, 200, 639):
        k2 = k.copy()
        k2[:, :, spike_at] = round_to(4.0 * np.sign(q[:, :, -1]) , "bf16")   # aligns with the last query
        for causal in (False, True):
            want = sdpa_ref(q, k2, v, causal=causal)
            o = run_fwd(sfa, q, k2, v, "bf16", causal)
            np.testing.assert_allclose(o, want, atol=1.6e-2, rtol=1.6e-2)


@pytest.mark.parametrize("impl,D", NON_BASELINE)
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_prefill_extreme_logits(sfa, knobs, impl, D, dtype):
    """Scores two orders of magnitude beyond N(0,1) data (row maxima jump by hundreds of log2 units from
    tile to tile, softmax nearly one-hot): the reference max must move BEFORE any exponential is taken.
    No inf/NaN, and the output matches the fp64 oracle."""
    select_impl(knobs, impl)
    rng = np.random.default_rng(21)
    B, H, S = 1, 2, 700
    q = round_to(6.0 * rng.standard_normal((B, H, S, D)), dtype)
    k = round_to(6.0 * rng.standard_normal((B, H, S, D)), dtype)
    k[:, :, 400:] *= 3.0                                    # later tiles dominate by a wide margin
    k = round_to(k, dtype)
    v = round_to(rng.standard_normal((B, H, S, D)), dtype)
    for causal in (False, True):
        want = sdpa_ref(q, k, v, causal=causal)
        o = run_fwd(sfa, q, k, v, dtype, causal)
        assert np.isfinite(o).all()
        # one-hot-ish rows: the output is (nearly) a single V row; a near-tie between two keys can flip
        bad = np.abs(o - want) > (TOL[dtype] + TOL[dtype] * np.abs(want))
        if impl.startswith("prescaled"):
            # Q*scale rounded to 16 bit: the score error scales with the logits (hundreds here) -- this
            # is why the flavour is opt-in.  Still finite, still mostly right.
            assert bad.mean() < 0.1, bad.mean()
        else:
            assert bad.mean() < 2e-3, bad.mean()


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_prefill_flavours_and_geometries(sfa, knobs, causal, dtype):
    """Within a numeric flavour the 256-row (1 or 2 pairs per workgroup) and the 128-row kernels are
    bit-identical (same MFMA order per row, same per-32-row rescale decisions); the exact flavour is the default, the prescaled one needs fast_scale=True and is
    never used when the log-sum-exp is returned."""
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, H, S, D = 1, 3, 1100, 128
    q, k, v = (torch.randn(B, H, S, D, device=dev).to(TDT[dtype]) for _ in range(3))

    def run(impl, lse, fast=False):
        select_impl(knobs, impl)
        r = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=lse, fast_scale=fast)
        torch.cuda.synchronize()
        return r[0] if lse else r

    exact = run("rows256", True)
    assert torch.equal(exact, run("rows256x2", True))
    assert torch.equal(exact, run("rows128", True))
    pre = run("prescaled256", False)
    assert torch.equal(pre, run("prescaled256x2", False))
    assert torch.equal(pre, run("prescaled128", False))
    assert torch.equal(run("auto", True), exact) and torch.equal(run("auto", False), exact)   # default: exact
    assert torch.equal(run("auto", False, fast=True), pre)          # opted in, output only -> prescaled Q
    assert torch.equal(run("auto", True, fast=True), exact)         # LSE requested -> exact regardless
    # the 4-wave kernel sums each row in a different order (one accumulator per query block, elements in PV
    # order): the same flavours to a 16-bit rounding flip, not to the bit
    tol = TOL[dtype]
    np.testing.assert_allclose(run("w4", True).float().cpu().numpy(), exact.float().cpu().numpy(), atol=tol / 2, rtol=tol / 2)
    np.testing.assert_allclose(run("prescaled_w4", False).float().cpu().numpy(), pre.float().cpu().numpy(), atol=tol / 2, rtol=tol / 2)
    # the two flavours differ by 16-bit rounding flips only
    np.testing.assert_allclose(pre.float().cpu().numpy(), exact.float().cpu().numpy(), atol=tol, rtol=tol)


def test_prefill_argument_errors(sfa):
    dev = torch.device("cuda:0")
    q = torch.zeros(1, 2, 16, 128, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="float16 or bfloat16"):
        sfa.flash_attn_fwd(q.float(), q.float(), q.float())
    with pytest.raises(RuntimeError, match="dtype"):
        sfa.flash_attn_fwd(q, q.half(), q)
    with pytest.raises(RuntimeError, match="head_dim"):
        z = torch.zeros(1, 2, 16, 96, device=dev, dtype=torch.bfloat16)
        sfa.flash_attn_fwd(z, z, z)
    with pytest.raises(RuntimeError, match="heads_q"):
        sfa.flash_attn_fwd(q.expand(1, 2, 16, 128)[:, :, :, :].repeat(1, 2, 1, 1)[:, :3], q, q)
