#!/usr/bin/env python3
"""Small-batch decode latency vs split count (the reference's own use case: B=2, H=32, fp16):
per-call time of flash_decode, HIP events over 200 back-to-back calls, auto split marked."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
from starflashattention_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
for (B, H, M, D) in ((1, 32, 8192, 128), (2, 32, 8192, 128), (2, 32, 2048, 128), (8, 32, 4096, 128), (16, 32, 8192, 128),
                     (32, 32, 2048, 128), (64, 32, 8192, 128)):
    kc = torch.randn(B, 1, M, H, D, device=dev).half()
    vc = torch.randn(B, 1, M, H, D, device=dev).half()
    qkv = torch.randn(B, 3, H, D, device=dev).half()
    o = torch.empty(B, H, D, device=dev, dtype=torch.float16)
    sl = torch.full((B,), M - 1, dtype=torch.int32, device=dev)
    z = torch.zeros(0, dtype=torch.float16, device=dev)
    auto = lib.sfa_decode_auto_splits(B, H, D, M)
    row = []
    for S in (1, 2, 4, 8, 16, 32, 64):
        run = lambda: sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, 1, 0, num_splits=S)
        for _ in range(20):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        row.append(f"S={S}{'*' if S == auto else ''}: {us:6.1f}us")
    gb = 2 * B * M * H * D * 2 / 1e9
    print(f"B={B} H={H} M={M}: KV {gb * 1e3:.0f} MB (floor {gb / 6.5e3 * 1e6:.0f}us at 6.5 TB/s) | " + "  ".join(row), flush=True)
