#!/usr/bin/env python3
"""A/B of the decode kernel at BASELINE.json configs[3] (B=256 Sq=1 Sk=8192 H=32 D=128 bf16):
load policy (knob decode_nt: 0 default, 1 non-temporal) x cache layout (LAYOUTS=blmhd,blhmd,paged:16).
Usage: [LAYOUTS=blmhd,blhmd] decode_ab.py [nt ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa

variants = [int(a) for a in sys.argv[1:]] or [0, 1]
layouts = os.environ.get("LAYOUTS", "blmhd").split(",")
B, H, Sk, D = 256, 32, 8192, int(os.environ.get("HD", "128"))
if os.environ.get("MFMA"):                   # grouped queries: 0 forces the VALU kernel, 1 the matrix-core kernel
    sfa.debug_set("decode_gqa_mfma", int(os.environ["MFMA"]))
GQA = int(os.environ.get("GQA", "1"))       # query heads per kv head (extension): Hq = 32, Hkv = 32 / GQA
HQ, H = H, H // GQA                         # from here on H = kv heads
dev = torch.device("cuda:0")
kc = torch.empty((B, 1, Sk, H, D), dtype=torch.bfloat16, device=dev)
vc = torch.empty_like(kc)
for t in (kc, vc):
    flat = t.view(-1)
    for i in range(0, flat.numel(), 1 << 28):
        flat[i:i + (1 << 28)].normal_()
qkv = torch.randn((B, 3, H, D) if GQA == 1 else (B, HQ + 2 * H, D), device=dev).bfloat16()
sl = torch.full((B,), Sk - 1, dtype=torch.int32, device=dev)
z = torch.zeros(0, dtype=torch.bfloat16, device=dev)
nbytes = 2.0 * B * Sk * H * D * 2 + 2 * B * (HQ + H) * D * 2
for rep in range(2):
    for layout in layouts:
        # the same bytes re-interpreted in the other layout: random data either way
        kw = {"kv_layout": layout.split(":")[0]}
        if GQA > 1:
            kw["num_heads_kv"] = H
        if layout.startswith("paged"):          # paged:PS -- pool of B*Sk/PS pages, randomly assigned
            ps = int(layout.split(":")[1])
            shape = (B * Sk // ps, 1, ps, H, D)
            kw["block_table"] = torch.randperm(B * Sk // ps, device=dev).to(torch.int32).view(B, Sk // ps)
        else:
            shape = (B, 1, Sk, H, D) if layout == "blmhd" else (B, 1, H, Sk, D)
        k, v = kc.view(shape), vc.view(shape)
        for nt in variants:
            sfa.debug_set("decode_nt", nt)
            o = torch.empty((B, HQ, D), dtype=torch.bfloat16, device=dev)
            run = lambda: sfa.flash_decode(qkv, z, z, z, k, v, sl, o, B, Sk, HQ, D, D, Sk, 1, 0, **kw)
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(6):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            print(f"D {D} GQA {GQA} mfma {os.environ.get('MFMA', 'auto')} layout {layout} nt {nt}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s", flush=True)
