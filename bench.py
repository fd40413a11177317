#!/usr/bin/env python3
"""Headline benchmark: attention forward TFLOPS/GPU, bf16, seqlen=4096, hdim=128, causal
(BASELINE.json configs[2], the Llama-3-8B shape B=16 H=32), on synthetic N(0,1) tensors.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (WORLD_SIZE unset) and --gpus N > 1, it launches itself: N fresh child
processes through torch.distributed.run, before this process has touched a GPU; rank 0's JSON line is
the output.  One process per GPU.  The batch x heads axis shards embarrassingly: every rank owns its own
B=16 shard (weak scaling) and there is NO collective on the data path -- torch.distributed (RCCL)
is used only for the two timing barriers and the max-over-ranks reduction of the elapsed time.

A "step" = one launch of the prefill kernel over the rank's whole shard, inputs resident in HBM.
FLOPs follow the FlashAttention convention: 4*B*H*Sq*Sk*D, halved for causal.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0         # HBM3E spec

WORKLOAD = dict(batch=16, heads=32, seqlen=4096, head_dim=128, causal=True)
DECODE = dict(batch=256, heads=32, seqlen_k=8192, head_dim=128)     # BASELINE.json configs[3]


def attn_flops(B, H, Sq, Sk, D, causal):
    f = 4.0 * B * H * Sq * Sk * D
    return f / 2 if causal else f


def cpu_baseline(torch, seconds_budget=20.0):
    """PyTorch eager SDPA on the host cores (the north_star's stated CPU baseline), fp32, same
    seqlen/hdim/causal as the workload on a bounded sample of (batch, head) pairs."""
    from oracle import sdpa_torch_cpu          # cpu_baseline leg: allowed to use oracle/
    S, D = WORKLOAD["seqlen"], WORKLOAD["head_dim"]
    H = 8
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((1, H, S, D), generator=g) for _ in range(3))
    sdpa_torch_cpu(q, k, v, causal=True)       # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < seconds_budget / 2 and len(times) < 10):
        t0 = time.perf_counter()
        sdpa_torch_cpu(q, k, v, causal=True)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(attn_flops(1, H, S, S, D, True) / med / 1e12, 4), "unit": "TFLOPS",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"F.scaled_dot_product_attention fp32 is_causal, B=1 H={H} S={S} D={D} "
                      f"(1/{WORKLOAD['batch'] * WORKLOAD['heads'] // H} of the GPU step), "
                      f"median of {len(times)} runs, {med * 1e3:.0f} ms each"}


def bench_decode(torch, sfa, steps, warmup):
    """BASELINE.json configs[3]: B=256 Sq=1 Sk=8192 H=32 D=128 bf16 decode, HBM-bound.
    Algorithmic bytes = K + V rows read once + qkv + o."""
    dev = torch.device("cuda", torch.cuda.current_device())
    B, H, Sk, D = DECODE["batch"], DECODE["heads"], DECODE["seqlen_k"], DECODE["head_dim"]
    M = Sk
    free, _ = torch.cuda.mem_get_info()
    need = 2 * B * M * H * D * 2
    if free < need + (4 << 30):
        return {"skipped": f"needs {need >> 30} GiB"}
    kc = torch.empty((B, 1, M, H, D), dtype=torch.bfloat16, device=dev)
    vc = torch.empty_like(kc)
    # fill on the device, chunked (a 16 GiB normal_() temp would double the footprint)
    for t in (kc, vc):
        flat = t.view(-1)
        step = 1 << 28
        for i in range(0, flat.numel(), step):
            flat[i:i + step].normal_()
    qkv = torch.randn((B, 3, H, D), device=dev).bfloat16()
    o = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
    sl = torch.full((B,), Sk - 1, dtype=torch.int32, device=dev)
    z = torch.zeros(0, dtype=torch.bfloat16, device=dev)
    run = lambda: sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, 1, 0)
    for _ in range(warmup):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = 2.0 * B * Sk * H * D * 2 + (3 + 1) * B * H * D * 2
    gbps = nbytes / (ms * 1e-3) / 1e9
    # the opt-in head-major cache layout: same bytes viewed as [B, L, H, M, D]
    kh, vh = kc.view(B, 1, H, M, D), vc.view(B, 1, H, M, D)
    run_h = lambda: sfa.flash_decode(qkv, z, z, z, kh, vh, sl, o, B, M, H, D, D, M, 1, 0, kv_layout="blhmd")
    run_h()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run_h()
    e1.record()
    torch.cuda.synchronize()
    gbps_h = nbytes / (e0.elapsed_time(e1) / steps * 1e-3) / 1e9
    del kc, vc, kh, vh
    torch.cuda.empty_cache()
    return {"workload": "decode B=256 Sq=1 Sk=8192 H=32 D=128 bf16 (fused RoPE+append)",
            "ms_per_step": round(ms, 4), "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS,
            "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4), "bound": "hbm",
            "algorithmic_bytes": nbytes,
            "head_major_layout": {"achieved": round(gbps_h, 1), "frac": round(gbps_h / PEAK_HBM_GBPS, 4)}}


def bench_config5_shard(torch, sfa, steps):
    """BASELINE.json configs[4]'s per-GPU shard (batch 128 over 8 GPUs = 16 per GPU): S=8192 non-causal."""
    ms, tf, name = time_prefill(torch, sfa, 16, 32, 8192, 128, False, steps)
    return {"workload": "prefill fwd full, B=16 H=32 S=8192 D=128 (BASELINE.json configs[4], one GPU's shard)",
            "kernel": name, "ms_per_step": round(ms, 4), "tflops": round(tf, 2), "frac_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4)}


def time_prefill(torch, sfa, B, H, S, D, causal, steps, warmup=3, Hkv=None, seed=77):
    """One prefill shape through the library's own kernel choice: (ms per launch, TFLOPS, kernel name)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(seed)
    q = torch.randn((B, H, S, D), generator=g, device=dev, dtype=torch.float32).bfloat16()
    k, v = (torch.randn((B, Hkv or H, S, D), generator=g, device=dev, dtype=torch.float32).bfloat16() for _ in range(2))
    out = torch.empty_like(q)
    for _ in range(warmup):
        sfa.flash_attn_fwd(q, k, v, causal=causal, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        sfa.flash_attn_fwd(q, k, v, causal=causal, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    name = sfa.last_prefill_kernel()
    del q, k, v, out
    torch.cuda.empty_cache()
    return ms, attn_flops(B, H, S, S, D, causal) / (ms * 1e-3) / 1e12, name


def bench_other_prefill(torch, sfa, steps):
    """The kernels the headline does not exercise (round-2 verdict item 5), never `value`:
    BASELINE.json configs[1] (head_dim 64, the 8-wave / 128-row kernels) and head_dim 256."""
    out = {}
    for key, (B, H, S, D, causal, what) in {
            "config2_bringup": (8, 16, 1024, 64, False, "prefill fwd full, B=8 H=16 S=1024 D=64 (BASELINE.json configs[1])"),
            "prefill_d256": (8, 16, 4096, 256, True, "prefill fwd causal, B=8 H=16 S=4096 D=256"),
    }.items():
        try:
            ms, tf, name = time_prefill(torch, sfa, B, H, S, D, causal, max(steps, 20) if S <= 1024 else steps)
            out[key] = {"workload": what, "kernel": name, "ms_per_step": round(ms, 4), "achieved": round(tf, 2),
                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "bound": "mfma"}
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    return out


def bench_decode_gqa(torch, sfa, steps):
    """Grouped-query decode, B=256 Hq=32 Hkv=4 Sk=8192 D=128 bf16: the cache is streamed once per KV head
    (decode_gqa_mfma_kernel); algorithmic bytes = K + V rows once + qkv + o."""
    dev = torch.device("cuda", torch.cuda.current_device())
    B, H, Hkv, Sk, D = 256, 32, 4, 8192, 128
    M = Sk
    kc = torch.randn((B, 1, M, Hkv, D), device=dev, dtype=torch.float32).bfloat16()
    vc = torch.randn((B, 1, M, Hkv, D), device=dev, dtype=torch.float32).bfloat16()
    qkv = torch.randn((B, H + 2 * Hkv, D), device=dev).bfloat16()
    o = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
    sl = torch.full((B,), Sk - 1, dtype=torch.int32, device=dev)
    z = torch.zeros(0, dtype=torch.bfloat16, device=dev)
    run = lambda: sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, M, H, D, D, M, 1, 0, num_heads_kv=Hkv)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = 2.0 * B * Sk * Hkv * D * 2 + (H + 2 * Hkv + H) * B * D * 2
    gbps = nbytes / (ms * 1e-3) / 1e9
    del kc, vc
    torch.cuda.empty_cache()
    return {"workload": "decode B=256 Sq=1 Sk=8192 Hq=32 Hkv=4 D=128 bf16 (grouped queries on the matrix cores)",
            "ms_per_step": round(ms, 4), "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": round(gbps / PEAK_HBM_GBPS, 4), "bound": "hbm", "algorithmic_bytes": nbytes}


def bench_gemm_calibration(torch):
    """What the matrix cores of THIS device sustain on a plain library GEMM (hipBLASLt through torch.matmul) with the
    operand data of the attention benchmark (randn, bf16): the device clocks down under dense MFMA work on real data
    (DESIGN.md 5.2 "Power"), so this -- not the nominal 2.5 PFLOP/s `roofline.peak` must use -- is the practical
    ceiling next to which the attention figure reads.  A yardstick only: nothing of the product runs through it."""
    dev = torch.device("cuda", torch.cuda.current_device())
    M = 8192
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn((M, M), generator=g, device=dev).bfloat16()
    bt = torch.randn((M, M), generator=g, device=dev).bfloat16().t()       # "NT": the library's fastest layout here
    for _ in range(5):
        torch.matmul(a, bt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        torch.matmul(a, bt)
    e1.record()
    torch.cuda.synchronize()
    tf = 2.0 * M ** 3 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12
    return {"workload": "torch.matmul bf16 8192^3, randn operands, B transposed (hipBLASLt)", "tflops": round(tf, 1),
            "frac_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4),
            "note": "library GEMM on the same device and data distribution: the power-capped MFMA rate, not a target"}


def measure_traffic(timeout_s=150):
    """--measure-traffic: HBM bytes per launch of the headline kernel, measured NOW -- two rocprofv3 child processes
    (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) over tools/prefill_once.py, which
    launches the same kernel on the same shape.  Children, never an exec; counters with --kernel-trace only.  Returns
    (bytes, source) or (None, reason).  FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="sfa_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", N="6")
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                                sys.executable, os.path.join(ROOT, "tools", "prefill_once.py")],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}"
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "prefill_w4_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, f"no {counter} rows for the prefill kernel"
            vals[counter] = sum(got[1:]) / max(1, len(got) - 1) if len(got) > 1 else got[0]     # first launch: cold
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    total = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    return total, "measured in this run: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over tools/prefill_once.py (same kernel, same shape), per launch"


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start N fresh ranks (torch.distributed.run) and relay their output.
    Nothing in THIS process has touched a GPU yet (torch is not even imported), and it is never replaced by
    exec: the ranks are children, this process exits with their code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults long enough to time the steady state: the first ~20 launches after start-up run up to
    # 6 % slower (10 steps after 3 warm-ups: 997 TFLOPS; 100 after 20: 1058 on the same device)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the configs[4] shard under --gpus N > 1")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="measure roofline.traffic now (two rocprofv3 --pmc child runs, ~40 s) instead of citing profiles/")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch
    import starflashattention_amd as sfa
    sfa._lib.load()                      # fail loudly if the HIP library is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE={world})")
    # one rank per GPU.  SFA_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks
    # (ranks then share devices): it exercises this launch/timing path, not RCCL.
    backend = os.environ.get("SFA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL; barriers + one scalar reduction only
        else:
            dist.init_process_group(backend)

    B, H, S, D = (WORKLOAD[k] for k in ("batch", "heads", "seqlen", "head_dim"))
    causal = WORKLOAD["causal"]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)      # every rank: its own shard
    q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev, dtype=torch.float32).bfloat16()
               for _ in range(3))
    out = torch.empty_like(q)

    def step():
        sfa.flash_attn_fwd(q, k, v, causal=causal, out=out)

    for _ in range(args.warmup):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()                                  # same stream the kernel is launched on
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps          # avg launch duration (back-to-back launches)
    kernel_name = sfa.last_prefill_kernel()               # what the dispatcher launched, not a string literal
    from starflashattention_amd.sharding import aggregate_throughput, gather_over_ranks, max_over_ranks
    per_rank_ms = gather_over_ranks(elapsed / args.steps * 1e3, dist, dev)     # (devices of one node differ by several %)
    elapsed = max_over_ranks(elapsed, dist, dev)          # the job is as slow as its slowest rank
    # BASELINE.json configs[4] is the multi-GPU workload proper (B=128 S=8192 non-causal over 8 GPUs = B=16 per GPU):
    # under --gpus N every rank also times its shard of it; still no data-path collective
    c5 = None
    if world > 1 and not args.no_config5:
        c5_steps = min(10, max(3, args.steps // 4))
        if dist is not None:
            dist.barrier()
        c5_ms, _, c5_name = time_prefill(torch, sfa, 16, 32, 8192, 128, False, c5_steps, seed=77 + rank)
        c5_all = gather_over_ranks(c5_ms, dist, dev)
        c5 = (c5_all, c5_name)

    flops_step = attn_flops(B, H, S, S, D, causal)        # per rank
    total_tflops = aggregate_throughput(flops_step, args.steps, elapsed, world) / 1e12
    kern_tflops = flops_step / (kernel_ms * 1e-3) / 1e12

    if rank == 0:
        traffic, traffic_src = None, None
        for name in ("r03_hbm_traffic.json", "r02_hbm_traffic.json"):   # committed PMC measurement of this same command (tools/profile_bench.sh)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic = json.load(f)["prefill_kernel"]["total_bytes"]
                    traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
                break
            except Exception:
                pass
        if args.measure_traffic and world == 1:
            try:
                t_now, why = measure_traffic()
            except Exception as e:
                t_now, why = None, repr(e)[:200]
            if t_now is not None:
                traffic, traffic_src = t_now, why
            else:
                traffic_src = f"{traffic_src}; --measure-traffic failed: {why}"
        rec = {
            "metric": "attention fwd TFLOPS/GPU (% MFMA peak), bf16 seqlen=4096 hdim=128",
            "value": round(total_tflops, 2), "unit": "TFLOPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "prefill fwd causal, B=16 H=32 S=4096 D=128 per GPU "
                                   "(BASELINE.json configs[2]); batch-sharded, no collectives",
                       "global_batch": B * world, "seq_len": S, "heads": H, "head_dim": D,
                       "causal": causal, "parallelism": f"batch-shard x{world}"},
            "tflops_per_gpu": round(total_tflops / world, 2),
            "frac_mfma_peak": round(total_tflops / world / PEAK_BF16_TFLOPS, 4),
            "per_rank_ms": {"headline": [round(x, 4) for x in per_rank_ms]},
            "roofline": {"bound": "mfma", "kernel": kernel_name,
                         "achieved": round(kern_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(kern_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_flops": flops_step},
        }
        if c5 is not None:
            c5_all, c5_name = c5
            f5 = attn_flops(16, 32, 8192, 8192, 128, False)
            tot = world * f5 / (max(c5_all) * 1e-3) / 1e12          # all ranks' work over the slowest rank's time
            rec["config5"] = {"workload": "prefill fwd full, B=16 H=32 S=8192 D=128 per GPU = BASELINE.json configs[4] "
                                          f"(B=128 over 8 GPUs) at {world} of its 8 shards; batch-sharded, no collectives",
                              "kernel": c5_name, "tflops_total": round(tot, 2), "tflops_per_gpu": round(tot / world, 2),
                              "frac_mfma_peak": round(tot / world / PEAK_BF16_TFLOPS, 4)}
            rec["per_rank_ms"]["config5"] = [round(x, 4) for x in c5_all]
        if world == 1:
            # supplementary, NOT the headline: the opt-in prescaled-Q kernels (fast_scale=True)
            try:
                fs = lambda: sfa.flash_attn_fwd(q, k, v, causal=causal, out=out, fast_scale=True)
                for _ in range(10):
                    fs()
                torch.cuda.synchronize()
                e0.record()
                nfs = max(10, args.steps // 4)
                for _ in range(nfs):
                    fs()
                e1.record()
                torch.cuda.synchronize()
                rec["fast_scale_variant"] = {
                    "tflops": round(flops_step / (e0.elapsed_time(e1) / nfs * 1e-3) / 1e12, 2),
                    "note": "opt-in fast_scale=True (Q*scale rounded to bf16 once; not the default path)"}
            except Exception as e:
                rec["fast_scale_variant"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_decode:
            try:
                del q, k, v, out
                torch.cuda.empty_cache()
                rec["config5_shard"] = bench_config5_shard(torch, sfa, min(20, max(3, args.steps // 4)))
                torch.cuda.empty_cache()
            except Exception as e:
                rec["config5_shard"] = {"error": repr(e)[:200]}
            rec.update(bench_other_prefill(torch, sfa, min(20, max(3, args.steps // 4))))
            try:
                rec["decode_gqa"] = bench_decode_gqa(torch, sfa, min(30, max(3, args.steps // 4)))
            except Exception as e:
                rec["decode_gqa"] = {"error": repr(e)[:200]}
            try:
                rec["gemm_calibration"] = bench_gemm_calibration(torch)
                torch.cuda.empty_cache()
            except Exception as e:
                rec["gemm_calibration"] = {"error": repr(e)[:200]}
            try:
                rec["decode_roofline"] = bench_decode(torch, sfa, min(50, max(3, args.steps // 2)), 3)
            except Exception as e:                      # the headline number must still print
                rec["decode_roofline"] = {"error": repr(e)[:200]}
            # the headline workload once more, now that the device has been busy for seconds: with few --steps / --warmup the
            # timed region above sits in the first ~50 ms after start-up, where launches run up to 6 % slower (never `value`)
            try:
                ms, tf, name = time_prefill(torch, sfa, B, H, S, D, causal, 100, warmup=20, seed=1234)
                rec["headline_warm"] = {"workload": "the headline workload again at the end of this run, 100 steps after 20 warm-ups",
                                        "kernel": name, "ms_per_step": round(ms, 4), "tflops": round(tf, 2),
                                        "frac_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4)}
            except Exception as e:
                rec["headline_warm"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(torch)
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
