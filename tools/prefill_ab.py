#!/usr/bin/env python3
"""In-process A/B of the prefill kernels at the headline shape (interleaved rounds; impl numbers:
prefill_dispatch.hip -- -1 auto, 1 8-wave (exact scale), 3 prescaled Q, 10 exact forced, 20-22 128-row,
40-42 the 4-wave persistent kernel; 0 baseline and 30-32 16x16x32 need the A/B library:
SFA_LIB_PATH=starflashattention_amd/lib/libStarFlashAttention_ab.so;
cdna_hip_programming.md rule 24).  usage: python tools/prefill_ab.py [impl ...] [--noncausal] [--d64 | --d256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa

impls = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [1, 40]
causal = "--noncausal" not in sys.argv
D = 64 if "--d64" in sys.argv else 256 if "--d256" in sys.argv else 128
B, H, S = 16, 32, 4096
for a in sys.argv[1:]:
    if a.startswith("--shape="):          # --shape=B,H,S
        B, H, S = (int(x) for x in a.split("=")[1].split(","))
Sk = S
for a in sys.argv[1:]:
    if a.startswith("--sk="):             # keys per sequence when it is not the query count (causal: the mask is bottom-right aligned)
        Sk = int(a.split("=")[1])
fp16 = "--fp16" in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn((B, H, n, D), generator=g, device=dev).to(torch.float16 if fp16 else torch.bfloat16) for n in (S, Sk, Sk))
if "--kvshared" in sys.argv:       # every (batch, head) reads the SAME K/V (stride 0): all re-reads hit L2
    k, v = k[:1, :1].expand(B, H, Sk, D), v[:1, :1].expand(B, H, Sk, D)
outs = {}
flops = 4.0 * B * H * S * S * D / (2 if causal else 1)
if Sk != S:                               # visible (query, key) pairs: row i sees keys 0 .. i + Sk - S
    vis = sum(min(Sk, max(0, i + Sk - S + 1)) for i in range(S)) if causal else S * Sk
    flops = 4.0 * B * H * vis * D
def run(impl, n):
    sfa.debug_set("prefill_impl", impl)
    o = torch.empty_like(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        sfa.flash_attn_fwd(q, k, v, causal=causal, out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, o
for i in impls:
    run(i, 3)
res = {i: [] for i in impls}
for rnd in range(5):
    for i in impls:
        ms, o = run(i, 5)
        res[i].append(ms); outs[i] = o
ref = outs[impls[0]].float()
for i in impls:
    ms = sorted(res[i]); med = ms[len(ms)//2]
    err = (outs[i].float() - ref).abs().max().item()
    print(f"[B={B} H={H} S={S}{'' if Sk == S else f' Sk={Sk}'} D={D} {'fp16' if fp16 else 'bf16'} {'causal' if causal else 'full'}] impl {i}: median {med:.3f} ms  min {ms[0]:.3f} ms  {flops/med/1e9:.1f} TFLOPS (best {flops/ms[0]/1e9:.1f})  max|diff vs impl {impls[0]}| = {err:.4g}", flush=True)
