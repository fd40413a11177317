#!/usr/bin/env python3
"""A few launches of the headline prefill shape (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
sfa.debug_set("prefill_impl", int(os.environ.get("IMPL", "-1")))
B, H, S, D = 16, 32, 4096, int(os.environ.get("HD", "128"))
if os.environ.get("SHAPE"):            # SHAPE=B,H,S
    B, H, S = (int(x) for x in os.environ["SHAPE"].split(","))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).bfloat16() for _ in range(3))
o = torch.empty_like(q)
for _ in range(int(os.environ.get("N", "4"))):
    sfa.flash_attn_fwd(q, k, v, causal=os.environ.get("CAUSAL", "1") == "1", out=o)
torch.cuda.synchronize()
