"""Multi-GPU layout of the hot path: the (batch, head) axis shards embarrassingly.

Every (b, h) pair is independent (reference flash_attn.cu:566-567: one CTA column per b*H+h), so
GPU g of N owns a contiguous batch range and runs the same kernels on it; there is NO data-path
collective.  torch.distributed (RCCL on ROCm, gloo in the CPU tests) is used for exactly two things
in bench.py: the barriers around the timed region and the max-over-ranks of the elapsed time.
"""


def batch_shard(total_batch: int, world_size: int, rank: int):
    """Contiguous [start, stop) of the batch owned by `rank`; the first total % world ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    base, extra = divmod(total_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of one float (the whole job is as slow as its slowest rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, dist=None, device=None):
    """Every rank's value of one float, in rank order (one all_gather of a scalar: bench.py's per-rank times)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def aggregate_throughput(units_per_rank_per_step: float, steps: int, elapsed_max_s: float, world_size: int) -> float:
    """Whole-job units/s: all ranks' units divided by the slowest rank's time."""
    return world_size * units_per_rank_per_step * steps / elapsed_max_s
