#!/usr/bin/env python3
"""Diagnostic (prefill_impl 43: the q-tile stamping build of the round-3 4-wave kernel, A/B library only): where the
life of a q-tile goes, per wave of workgroup 8, over its first 16 q-tiles.  Stamps per q-tile: 0 top (in front of the Q
fetch), 1 behind the barrier, 2 behind the seam half-step and the previous q-tile's epilogue, 3 behind the full steps,
4 in front of the next top.  Cycles, clock and wall for each segment (cdna_hip_programming.md rule 28).
usage: SFA_LIB_PATH=.../libStarFlashAttention_ab.so python tools/w4_seam_stamps.py [--noncausal] [--shape=B,H,S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", 43)
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
st = lse.view(-1)[: 4 * 16 * 8 * 2].view(torch.int64).view(4, 16, 8).cpu().double()
names = ["Q fetch+barrier", "seam hstep+epi", "full steps", "last tile/idle"]
print("cycles per q-tile, mean over q-tiles 1..15:  " + "  ".join(f"{n:>16s}" for n in names) + "   exposed epilogue   per full step   clock")
for w in range(4):
    s = st[w, 1:]
    d = [(s[:, i + 1] - s[:, i]).mean().item() for i in range(4)]
    ntw = torch.tensor([int(x) & 0xffffffff for x in s[:, 5].tolist()]).double()
    per = ((s[:, 3] - s[:, 2]) / (ntw - 1).clamp(min=1)).mean().item()
    clk = ((s[:, 4] - s[:, 0]).sum() / s[:, 6].sum() / 10).item()
    print(f"wave {w}:                                      " + "  ".join(f"{x:16.0f}" for x in d) +
          f"   {s[:, 7].mean().item():12.0f}   {per:12.0f}   {clk:.3f} GHz")
print("wave 3, per q-tile:")
for it in range(16):
    s = st[3, it]
    ntw, nt = int(s[5].item()) & 0xffffffff, int(s[5].item()) >> 32
    n = max(1.0, ntw - 1)
    clk = ((s[4] - s[0]) / s[6] / 10).item() if s[6] > 0 else 0
    print(f"   q-tile {it:2d}: ntw/nt {ntw:3d}/{nt:3d}  fetch+barrier {(s[1] - s[0]).item():6.0f}  seam+epi {(s[2] - s[1]).item():6.0f} (epi {s[7].item():6.0f})  "
          f"per full step {((s[3] - s[2]) / n).item():7.0f}  last/idle {(s[4] - s[3]).item():7.0f}  whole {(s[4] - s[0]).item():8.0f}  {clk:.3f} GHz")
tot = (st[:, -1, 4] - st[:, 0, 0]).mean().item()
print(f"16 q-tiles: {tot:.0f} cycles per wave (top of the first to the end of the last)")
