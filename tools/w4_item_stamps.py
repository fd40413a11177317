#!/usr/bin/env python3
"""Diagnostic (prefill_impl 96, round 2 kernel, A/B library: the q-tile stamping build of the 4-wave kernel): where the life of a q-tile goes,
per wave of workgroup 8, over its first 16 q-tiles.  usage: [--noncausal]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
impl = 99 if "--qpre" in sys.argv else 96
for a in sys.argv[1:]:
    if a.startswith("--impl="):          # the stamping builds of the ablations (A/B library): 61 empty descriptors, 62 no DMA,
        impl = int(a.split("=")[1])      # 63 no softmax, 64 MFMAs only, 65 everything but the load instruction
sfa.debug_set("prefill_impl", impl)
B, H, S, D = 16, 32, 4096, 128
for a in sys.argv[1:]:
    if a.startswith("--shape="):
        B, H, S = (int(x) for x in a.split("=")[1].split(","))
causal = "--noncausal" not in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
if "--kvshared" in sys.argv:       # every (batch, head) reads the SAME K/V (stride 0): all re-reads hit L2
    k, v = k[:1, :1].expand(B, H, S, D), v[:1, :1].expand(B, H, S, D)
for _ in range(3):
    out, lse = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=True)
torch.cuda.synchronize()
st = lse.view(-1)[: 4 * 16 * 8 * 2].view(torch.int64).view(4, 16, 8).cpu().double()
names = ["Q wait", "first half-tile", "full steps", "tail + idle", "epilogue", "to next start"]
print("cycles per q-tile, mean over the 16 q-tiles:  " + "  ".join(f"{n:>15s}" for n in names) + "   per full step   ntw / nt (tile 0..3)")
for w in range(4):
    s = st[w]
    d = [(s[:, i + 1] - s[:, i]).mean().item() for i in range(5)]
    d.append((s[1:, 0] - s[:-1, 5]).mean().item())
    steps = (torch.tensor([int(x) & 0xffffffff for x in s[:, 6].tolist()]).double() - 1).clamp(min=1)
    per = ((s[:, 3] - s[:, 2]) / steps).mean().item()
    print(f"wave {w}:                                       " + "  ".join(f"{x:15.0f}" for x in d) + f"   {per:10.0f}      " +
          " ".join(f"{int(a) & 0xffffffff}/{int(a) >> 32}" for a in s[:4, 6].tolist()) +
          f"   clock {((s[:, 5] - s[:, 0]).sum() / s[:, 7].sum() / 10).item():.3f} GHz")
print("wave 3, per q-tile: tiles, cycles per full step, Q wait, epilogue")
for it in range(16):
    s = st[3, it]
    ntw = int(s[6].item()) & 0xffffffff
    n = max(1.0, ntw - 1)
    print(f"   q-tile {it:2d}: ntw {ntw:3d}  per step {((s[3] - s[2]) / n).item():7.0f}  Q wait {(s[1] - s[0]).item():6.0f}  "
          f"epilogue {(s[5] - s[4]).item():6.0f}  whole q-tile {(s[5] - s[0]).item():8.0f}")
tot = (st[:, -1, 5] - st[:, 0, 0]).mean().item()
print(f"16 q-tiles: {tot:.0f} cycles per wave")
