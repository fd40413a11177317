"""Launch attributes are per device (hipFuncSetAttribute(MaxDynamicSharedMemorySize) -- the persistent prefill kernels
ask for the CU's whole 160 KiB of LDS): the same process must be able to run every kernel family on a second device
after the first one.  Skipped on a one-GPU box."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    import starflashattention_amd as m
    m._lib.load()
    return m


def _prefill(sfa, dev, D, impl):
    g = torch.Generator(device="cpu").manual_seed(D)
    q, k, v = (torch.randn((1, 2, 384, D), generator=g).bfloat16().to(dev) for _ in range(3))
    sfa.debug_set("prefill_impl", impl)
    try:
        o = sfa.flash_attn_fwd(q, k, v, causal=True)
        torch.cuda.synchronize(dev)
    finally:
        sfa.debug_set("prefill_impl", -1)
    return o.float().cpu().numpy()


def _decode_gqa(sfa, dev):
    g = torch.Generator(device="cpu").manual_seed(3)
    B, Hkv, G, D, M, L = 2, 2, 8, 128, 128, 1
    H = Hkv * G
    qkv = torch.randn((B, H + 2 * Hkv, D), generator=g).bfloat16().to(dev)
    kc = torch.randn((B, L, M, Hkv, D), generator=g).bfloat16().to(dev)
    vc = torch.randn((B, L, M, Hkv, D), generator=g).bfloat16().to(dev)
    sl = torch.tensor([17, 100], dtype=torch.int32, device=dev)
    o = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
    z = torch.zeros(0, dtype=torch.bfloat16, device=dev)
    sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, L, 0, num_heads_kv=Hkv)
    sfa.check_decode_status(dev)
    return o.float().cpu().numpy()


def test_second_device_after_first(sfa):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d0, d1 = torch.device("cuda:0"), torch.device("cuda:1")
    for D, impl in ((128, 40), (128, 1), (256, -1), (64, -1)):      # 4-wave persistent, 8-wave, head_dim 256 persistent, head_dim 64
        a = _prefill(sfa, d0, D, impl)
        b = _prefill(sfa, d1, D, impl)
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(_decode_gqa(sfa, d0), _decode_gqa(sfa, d1))     # matrix-core grouped-query kernel (70 KiB LDS)
