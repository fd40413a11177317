#!/usr/bin/env python3
"""Calibration: what the matrix cores of THIS device sustain on a plain library GEMM (hipBLASLt through torch.matmul),
bf16, with the operand data the attention benchmark uses (randn) and with all-zero operands.  The difference is the
power cap: MI355X clocks down under dense MFMA work on real data (the 4-wave prefill kernel measures 1.79-1.83 GHz
against 2.4 GHz nominal with s_memtime / s_memrealtime, tools/w4_item_stamps.py)."""
import sys, torch
dev = torch.device("cuda:0")
def bench(a, b, n=20):
    for _ in range(3): torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): torch.matmul(a, b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (4096, 8192, 16384):
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((M, M), generator=g, device=dev).bfloat16()
    b = torch.randn((M, M), generator=g, device=dev).bfloat16()
    fl = 2.0 * M ** 3
    ms = bench(a, b)
    z = torch.zeros_like(a)
    msz = bench(z, z)
    bt = b.t().contiguous().t()          # "NT" layout, usually the library's fastest
    msn = bench(a, bt)
    print(f"bf16 GEMM {M}^3: randn {fl / ms / 1e9:7.1f} TFLOPS ({ms:.3f} ms)   randn, B transposed {fl / msn / 1e9:7.1f}   zeros {fl / msz / 1e9:7.1f} TFLOPS", flush=True)
