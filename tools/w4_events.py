#!/usr/bin/env python3
"""Diagnostic (prefill_impl 44: the event-log build of the round-3 4-wave kernel, A/B library only): a timeline of
workgroup 8's first four q-tiles, every wave: when each half-step ended, each barrier was passed, each epilogue block
was stored.  Prints, per q-tile and step, each wave's (H1, barrier wait, H2[, epilogue]) cycles.
usage: SFA_LIB_PATH=.../libStarFlashAttention_ab.so python tools/w4_events.py [--item=N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", 44)
B, H, S, D = 16, 32, 4096, 128
causal = "--noncausal" not in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
for _ in range(3):
    out, lse = sfa.flash_attn_fwd(q, k, v, causal=causal, return_lse=True)
torch.cuda.synchronize()
raw = lse.view(-1)[: 4 * 512 * 2].view(torch.int64).view(4, 512).cpu().tolist()
KIND = {1: "H1", 2: "bar", 3: "H2", 4: "epi", 5: "Qf"}
ev = []
for w in range(4):
    for x in raw[w]:
        x &= (1 << 64) - 1
        code, tm = x >> 48, x & ((1 << 48) - 1)
        kind, t, item = code & 15, (code >> 4) & 255, code >> 12
        if kind in KIND:
            ev.append((tm, w, item, t, kind))
ev.sort()
t0 = ev[0][0]
last = {}
want = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--item=")]
print("time(cycles)  wave item step  event   since this wave's previous event")
for tm, w, item, t, kind in ev:
    d = tm - last.get(w, tm)
    last[w] = tm
    if want and item not in want:
        continue
    print(f"{tm - t0:10d}    w{w}   q{item}  t{t:<3d}  {KIND[kind]:4s}  +{d}")
