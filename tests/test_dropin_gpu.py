"""GPU tests of the two drop-in surfaces that sit on top of the C ABI:
  * the pybind11 module `star_flash_attn` (reference src/flash_api.cpp:42-80): same call as the
    reference's example makes (examples/python/testFlashDecoder.py:124-126), positional and keyword;
  * the C++ template surface of src/flash_attn.h, through the torch-free harness
    examples/cpp/flash_decoder_harness.cc (scenario of the reference's examples/cpp/testFlashDecoder.cc).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, bf16bits_to_f32
from oracle import decode_ref, sdpa_ref, rotary_table_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import star_flash_attn          # built in-tree by __graft_entry__.build()
    return star_flash_attn


def test_reference_example_call_positional_and_keyword(ext, decode_golden):
    g = decode_golden
    dev = torch.device("cuda")
    qkv = torch.from_numpy(bf16bits_to_f32(g["qkv_bf16bits"])).half().to(dev)
    kc0 = torch.from_numpy(bf16bits_to_f32(g["k_cache_bf16bits"])).half().to(dev)
    vc0 = torch.from_numpy(bf16bits_to_f32(g["v_cache_bf16bits"])).half().to(dev)
    B, H, D, L, M = (int(x) for x in g["dims"])
    layer = int(g["idx_layer"])
    zeros = torch.zeros(H, D, dtype=torch.float16, device=dev)
    for case in (0, 3, 8):
        s = int(g["seq_lens"][case])
        seq_len = torch.full((B,), s, dtype=torch.int32, device=dev)
        kc, vc = kc0.clone(), vc0.clone()
        o = torch.zeros(B, H, D, dtype=torch.float16, device=dev)
        ret = ext.mha_fwd_cuda(qkv, zeros, zeros, zeros, kc, vc, seq_len, o, B, M, H, D, D, M, L, layer)
        assert ret.data_ptr() == o.data_ptr()
        kc2, vc2 = kc0.clone(), vc0.clone()
        o2 = torch.zeros_like(o)
        ext.mha_fwd_cuda(qkv=qkv, q_bias=zeros, k_bias=zeros, v_bias=zeros, k_cache_table=kc2,
                         v_cache_table=vc2, seq_len=seq_len, o=o2, batch_size=B, memory_max_len=M,
                         num_heads=H, head_dim=D, rotary_embedding_dim=D, max_input_length=M,
                         num_layer=L, idx_layer=layer)
        ext.check_errors()
        assert torch.equal(o, o2) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
        np.testing.assert_allclose(o.float().cpu().numpy(), g["o_f32"][case], atol=2e-3, rtol=2e-3)
        np.testing.assert_allclose(o.float().cpu().numpy(), g["o_f16"][case], atol=4e-3, rtol=4e-3)
        assert torch.equal(vc[:, layer, s], qkv[:, 2])


def test_extension_bias_bf16_and_errors(ext):
    dev = torch.device("cuda")
    rng = np.random.default_rng(2)
    B, H, D, L, M = 2, 3, 64, 2, 50
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).bfloat16().to(dev)
    qkv, kc, vc = mk(B, 3, H, D), mk(B, L, M, H, D), mk(B, L, M, H, D)
    bq, bk, bv = mk(H, D), mk(H, D), mk(H, D)
    lens = [49, 7]
    ref = decode_ref(qkv.float().cpu().numpy(), kc.float().cpu().numpy(), vc.float().cpu().numpy(), lens, 1, 32,
                     dtype="bf16", q_bias=bq.float().cpu().numpy(), k_bias=bk.float().cpu().numpy(),
                     v_bias=bv.float().cpu().numpy())
    o = torch.empty(B, H, D, dtype=torch.bfloat16, device=dev)
    ext.mha_fwd_cuda(qkv, bq, bk, bv, kc, vc, torch.tensor(lens, dtype=torch.int32, device=dev), o,
                     B, M, H, D, 32, M, L, 1)
    ext.check_errors()
    np.testing.assert_allclose(o.float().cpu().numpy(), ref["o"], atol=1.6e-2, rtol=1.6e-2)
    # argument errors raise, they do not print-and-continue as the reference does
    with pytest.raises(RuntimeError, match="shape"):
        ext.mha_fwd_cuda(qkv, bq, bk, bv, kc, vc, torch.tensor(lens, dtype=torch.int32, device=dev), o,
                         B, M + 1, H, D, 32, M, L, 1)
    with pytest.raises(RuntimeError, match="float16 or bfloat16"):
        ext.mha_fwd_cuda(qkv.float(), bq, bk, bv, kc, vc, torch.tensor(lens, dtype=torch.int32, device=dev), o,
                         B, M, H, D, 32, M, L, 1)
    # out-of-range seq_len: rejected on the device, reported by check_errors
    ext.mha_fwd_cuda(qkv, bq, bk, bv, kc, vc, torch.tensor([M, 3], dtype=torch.int32, device=dev), o,
                     B, M, H, D, 32, M, L, 1)
    with pytest.raises(RuntimeError, match="seq_len"):
        ext.check_errors()
    ext.check_errors()


def test_extension_prefill_and_rotary_table(ext):
    dev = torch.device("cuda")
    torch.manual_seed(1)
    q, k, v = (torch.randn(1, 4, 300, 128, device=dev).bfloat16() for _ in range(3))
    out, lse = ext.mha_fwd(q, k[:, :2], v[:, :2], causal=True, return_lse=True)
    want, lse_want = sdpa_ref(q.float().cpu().numpy(), k[:, :2].float().cpu().numpy(),
                              v[:, :2].float().cpu().numpy(), causal=True, return_lse=True)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, atol=1.6e-2, rtol=1.6e-2)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_want, atol=2e-3, rtol=2e-3)
    (out2,) = ext.mha_fwd(q, k[:, :2], v[:, :2], out=torch.empty_like(q), causal=True)
    assert torch.equal(out, out2)
    # opt-in prescaled-Q kernels: same result up to 16-bit rounding flips, checked against the oracle
    (out3,) = ext.mha_fwd(q, k[:, :2], v[:, :2], causal=True, fast_scale=True)
    np.testing.assert_allclose(out3.float().cpu().numpy(), want, atol=1.6e-2, rtol=1.6e-2)
    c, s = ext.compute_rotary_table(64, 128, torch.float16, dev)
    cr, sr = rotary_table_ref(64, 128, "fp16")
    assert np.max(np.abs(c.float().cpu().numpy() - cr)) <= 2.0 ** -10
    assert np.max(np.abs(s.float().cpu().numpy() - sr)) <= 2.0 ** -10


def test_python_example_runs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "python", "decode_dropin.py")],
                       cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max |star_flash_attn - torch|" in r.stdout


def test_cpp_harness_known_answers():
    exe = os.path.join(ROOT, "build", "flash_decoder_harness")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all cases ok" in r.stdout and "REJECTED (as it must be)" in r.stdout
    assert r.stdout.count(" ok") >= 6


def test_entry_points_are_graph_capturable():
    """the C-ABI entry points allocate nothing and never synchronise, so a decode
    step and a prefill call can be captured into a HIP graph (after a warm-up that sizes the workspace)
    and replayed on new data."""
    import starflashattention_amd as sfa
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    B, H, D, L, M = 4, 8, 128, 1, 512
    tdt = torch.float16
    qkv = torch.randn(B, 3, H, D, device=dev).to(tdt)
    kc = torch.randn(B, L, M, H, D, device=dev).to(tdt)
    vc = torch.randn(B, L, M, H, D, device=dev).to(tdt)
    sl = torch.tensor([5, 100, 300, 400], dtype=torch.int32, device=dev)
    o = torch.empty(B, H, D, device=dev, dtype=tdt)
    z = torch.zeros(0, dtype=tdt, device=dev)
    q = torch.randn(2, 4, 300, D, device=dev).to(tdt)
    k = torch.randn(2, 4, 300, D, device=dev).to(tdt)
    v = torch.randn(2, 4, 300, D, device=dev).to(tdt)
    po = torch.empty_like(q)

    def work():
        sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, L, 0, num_splits=3)
        sfa.flash_attn_fwd(q, k, v, causal=True, out=po)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):          # warm-up on the capture stream: sizes its workspace
        work()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        work()
    # new data in the same buffers, then replay
    kc0, vc0 = kc.clone(), vc.clone()
    qkv.copy_(torch.randn(B, 3, H, D, device=dev).to(tdt))
    q.copy_(torch.randn(2, 4, 300, D, device=dev).to(tdt))
    kc.copy_(kc0); vc.copy_(vc0)
    g.replay()
    torch.cuda.synchronize()
    o_g, po_g, kc_g = o.clone(), po.clone(), kc.clone()
    kc.copy_(kc0); vc.copy_(vc0)
    work()
    torch.cuda.synchronize()
    assert torch.equal(o, o_g) and torch.equal(po, po_g) and torch.equal(kc, kc_g)


@pytest.mark.parametrize("layout", ["blmhd", "blhmd", "paged"])
def test_prefill_and_decode_agree(layout):
    """The two hot paths against each other: the last row of a causal prefill over S tokens equals a
    decode step for token S-1 on a cache holding tokens 0..S-2 (RoPE off), in every cache layout; and the
    row the decode step appends is token S-1's K/V."""
    import starflashattention_amd as sfa
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    B, H, S, D, M, ps = 3, 4, 333, 128, 512, 16
    tdt = torch.bfloat16
    q, k, v = (torch.randn(B, H, S, D, device=dev).to(tdt) for _ in range(3))
    o_prefill = sfa.flash_attn_fwd(q, k, v, causal=True)
    kc = torch.zeros(B, 1, M, H, D, device=dev, dtype=tdt)
    vc = torch.zeros_like(kc)
    kc[:, 0, :S - 1] = k[:, :, :S - 1].transpose(1, 2)
    vc[:, 0, :S - 1] = v[:, :, :S - 1].transpose(1, 2)
    qkv = torch.stack([q[:, :, S - 1], k[:, :, S - 1], v[:, :, S - 1]], 1).contiguous()
    sl = torch.full((B,), S - 1, dtype=torch.int32, device=dev)
    o = torch.empty(B, H, D, device=dev, dtype=tdt)
    z = torch.zeros(0, dtype=tdt, device=dev)
    kw = {}
    if layout == "blhmd":
        kc, vc = (t.permute(0, 1, 3, 2, 4).contiguous() for t in (kc, vc))
        kw["kv_layout"] = "blhmd"
    elif layout == "paged":
        pps = M // ps
        table = torch.randperm(B * pps, device=dev).to(torch.int32).view(B, pps)
        def to_pool(c):
            pool = torch.zeros(B * pps, 1, ps, H, D, device=dev, dtype=tdt)
            pool[table.long().view(-1)] = c.view(B, 1, pps, ps, H, D).permute(0, 2, 1, 3, 4, 5).reshape(B * pps, 1, ps, H, D)
            return pool
        kc, vc = to_pool(kc), to_pool(vc)
        kw.update(kv_layout="paged", block_table=table)
    sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, 0, M, 1, 0, **kw)
    sfa.check_decode_status()
    np.testing.assert_allclose(o.float().cpu().numpy(), o_prefill[:, :, S - 1].float().cpu().numpy(),
                               atol=1.6e-2, rtol=1.6e-2)
    pos = S - 1
    for b in range(B):
        if layout == "blmhd":
            row_k = kc[b, 0, pos]
        elif layout == "blhmd":
            row_k = kc[b, 0, :, pos]
        else:
            row_k = kc[int(table[b, pos // ps]), 0, pos % ps]
        assert torch.equal(row_k, k[b, :, pos])
