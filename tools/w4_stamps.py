#!/usr/bin/env python3
"""Diagnostic (prefill_impl 84, A/B library: the stamping build of the 4-wave kernel): where a step's cycles go, per wave,
steps 8..15 of the first item of workgroup 8.  Stamps: 0 H1 start, 1 H1 end, 2 DMA wait over (vmcnt),
3 barrier passed, 4 H2 end.  usage: [--noncausal]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", 84)
B, H, S, D = 16, 32, 4096, 128
causal = "--noncausal" not in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
for _ in range(3):
    out, lse = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=True)
    lse_first = lse
torch.cuda.synchronize()
st = lse.view(-1)[: 2 * 4 * 8 * 8 * 2].view(torch.int64).view(2, 4, 8, 8).cpu().double()
for item, what in ((0, "steps 8..15 of the first q-tile"), (1, "steps 0..7 of the second q-tile")):
    print(f"{what}:  wave     H1   dma-wait  barrier     H2    step-total   (cycles, mean)")
    for w in range(4):
        s = st[item, w]
        d = s[:, 1:5] - s[:, 0:4]
        ok = ((d > 0) & (d < 1e6)).all(dim=1)       # entries of steps that were not full steps hold stale bits
        if ok.sum() < 2:
            continue
        s = s[ok]
        h1 = (s[:, 1] - s[:, 0]).mean().item(); dw = (s[:, 2] - s[:, 1]).mean().item()
        bw = (s[:, 3] - s[:, 2]).mean().item(); h2 = (s[:, 4] - s[:, 3]).mean().item()
        tot = (s[1:, 0] - s[:-1, 0]).mean().item()
        print(f"                               {w:3d}  {h1:7.0f}  {dw:8.0f}  {bw:7.0f}  {h2:7.0f}  {tot:10.0f}    per step (H1, wait, barrier, H2): " +
              " ".join("(%.0f %.0f %.0f %.0f)" % tuple((s[i, 1:5] - s[i, 0:4]).tolist()) for i in range(s.shape[0])))
