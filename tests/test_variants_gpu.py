"""The earlier prefill kernel generations (baseline, 16x16x32 MFMA) live in the A/B build of the library
only (build.py build_lib(variants=True) -> libStarFlashAttention_ab.so); launch_prefill never picks them.
Opt-in:  SFA_LIB_PATH=starflashattention_amd/lib/libStarFlashAttention_ab.so python -m pytest tests -m variants
(tools/gpu_ci.sh variants).  Without that library loaded these tests are skipped, not failed."""
import os

import numpy as np
import pytest
import torch

from oracle import round_to, sdpa_ref

pytestmark = [pytest.mark.gpu, pytest.mark.variants]
TOL = {"fp16": 2e-3, "bf16": 1.6e-2}
TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}
IMPLS = {"baseline": 0, "x16": 32, "prescaled_x16": 31}


@pytest.fixture(scope="module")
def sfa():
    if "_ab" not in os.path.basename(os.environ.get("SFA_LIB_PATH", "")):
        pytest.skip("needs the A/B library: SFA_LIB_PATH=.../libStarFlashAttention_ab.so")
    import starflashattention_amd as m
    m._lib.load()
    yield m
    m.debug_set("prefill_impl", -1)


@pytest.mark.parametrize("case", [(2, 3, 3, 64, 64, 128), (1, 2, 1, 257, 257, 128), (2, 2, 2, 513, 513, 64),
                                  (1, 2, 2, 100, 333, 128), (1, 2, 2, 1280, 1280, 128)],
                         ids=lambda c: "b%d_hq%d_hkv%d_sq%d_sk%d_d%d" % c)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("impl", list(IMPLS))
def test_variant_vs_oracle(sfa, case, causal, dtype, impl):
    sfa.debug_set("prefill_impl", IMPLS[impl])
    B, Hq, Hkv, Sq, Sk, D = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    q = round_to(rng.standard_normal((B, Hq, Sq, D)), dtype)
    k = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    v = round_to(rng.standard_normal((B, Hkv, Sk, D)), dtype)
    want = sdpa_ref(q, k, v, causal=causal)
    t = lambda x: torch.from_numpy(x).to(TDT[dtype]).to("cuda:0")
    o = sfa.flash_attn_fwd(t(q), t(k), t(v), causal=causal)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.float().cpu().numpy(), want, atol=TOL[dtype], rtol=TOL[dtype])
