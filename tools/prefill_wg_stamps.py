#!/usr/bin/env python3
"""Diagnostic (prefill_impl 4: the stamping build of the 8-wave kernel): where a workgroup's life goes.  Stamps: 0 workgroup start,
1 first q-tile's loop entry (end of the staging prologue), 2 end of the first q-tile's loop, 3 end of
the workgroup (second q-tile of the pair included).  usage: [--noncausal] [--shape=B,H,S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", 4)
B, H, S, D = 16, 32, 4096, 128
for a in sys.argv[1:]:
    if a.startswith("--shape="):
        B, H, S = (int(x) for x in a.split("=")[1].split(","))
causal = "--noncausal" not in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
for _ in range(3):
    out, lse = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=True)
torch.cuda.synchronize()
n = min(2048, 8 * ((B * H + 7) // 8) * ((((S + 255) // 256) + 1) // 2))
raw = lse.view(-1).view(torch.int64)[1024: 1024 + n * 8].view(n, 4, 2).cpu().double()
cyc, rt = raw[:, :, 0], raw[:, :, 1]          # shader cycles, 100 MHz ticks
ok = (rt[:, 3] > rt[:, 0]) & (rt[:, 0] > 0)
cyc, rt = cyc[ok], rt[ok]
us = lambda a, b_: ((rt[:, b_] - rt[:, a]) * 0.01)
print(f"[B={B} H={H} S={S} causal={causal}] workgroups sampled: {int(ok.sum())}")
print(f"  prologue  (start -> loop entry): {us(0,1).mean():7.2f} us   (min {us(0,1).min():.2f}, max {us(0,1).max():.2f})")
print(f"  first q-tile's steps           : {us(1,2).mean():7.2f} us")
print(f"  its epilogue + second q-tile   : {us(2,3).mean():7.2f} us   (min {us(2,3).min():.2f}, max {us(2,3).max():.2f})")
clk = ((cyc[:, 3] - cyc[:, 0]) / (rt[:, 3] - rt[:, 0]) * 100).median()
print(f"  shader clock inside the kernel : {clk:7.0f} MHz")
# gap: sort starts; for the wave of first-generation WGs (earliest 256 starts) find per-CU successor is unknown,
# so report the global picture: kernel span vs sum of WG lifetimes / 256 CUs
span = (rt[:, 3].max() - rt[:, 0].min()) * 0.01
life = ((rt[:, 3] - rt[:, 0]) * 0.01)
print(f"  sampled span {span:.1f} us; mean WG lifetime {life.mean():.2f} us; lifetimes*N/256 = {life.sum()/256:.1f} us "
      f"=> CU occupancy of the sampled window ~ {life.sum()/256/span*100:.0f} %")
