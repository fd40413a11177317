#!/usr/bin/env python3
"""Diagnostic: per-wave cycle shares of one workgroup's steps 8..15 (prefill_impl 4: the stamping build of the 8-wave kernel).
Stamps: 0 start of H1, 1 end of H1 (before barrier), 2 after barrier, 3 end of H2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", int(os.environ.get("STAMP_IMPL", "4")))
B, H, S, D = 16, 32, 4096, 128
causal = "--noncausal" not in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
for _ in range(3):
    out, lse = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=True)
torch.cuda.synchronize()
st = lse.view(-1)[: 8 * 8 * 4 * 2].view(torch.int64).view(8, 8, 4).cpu()
print("wave  H1      barrier-wait  H2      step-total   (cycles, mean over steps 8..15)")
for w in range(8):
    s = st[w].double()
    h1 = (s[:, 1] - s[:, 0]).mean().item(); bw = (s[:, 2] - s[:, 1]).mean().item(); h2 = (s[:, 3] - s[:, 2]).mean().item()
    tot = (s[1:, 0] - s[:-1, 0]).mean().item()
    print(f"{w:3d}  {h1:8.0f}  {bw:10.0f}  {h2:8.0f}  {tot:10.0f}")
skew = (st[:, :, 1].double() - st[:, :, 1].double().min(0).values).mean(1)
print("mean lateness at the barrier per wave (cycles):", [round(x) for x in skew.tolist()])
