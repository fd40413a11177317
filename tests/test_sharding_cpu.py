"""The N>1 path on CPU: batch sharding + the bench's timing protocol with world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from starflashattention_amd.sharding import aggregate_throughput, batch_shard, gather_over_ranks, max_over_ranks


def test_batch_shard_partitions_exactly():
    for total in (1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [batch_shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert batch_shard(128, 8, 3) == (48, 64)          # BASELINE config 5: 16 batches per GPU
    with pytest.raises(ValueError):
        batch_shard(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank owns its shard of a global batch and "computes" on it with no data exchange
        B, H = 6, 4
        full = torch.arange(B * H, dtype=torch.float32).reshape(B, H)
        lo, hi = batch_shard(B, world, rank)
        mine = full[lo:hi] * 2.0                            # stand-in for the per-shard kernel
        dist.barrier()
        elapsed = 0.010 * (rank + 1)                        # rank 1 is the slow one
        emax = max_over_ranks(elapsed, dist)
        every = gather_over_ranks(elapsed, dist)            # bench.py's per_rank_ms
        dist.barrier()
        gathered = [torch.zeros(B * H // world if B % world == 0 else 1) for _ in range(world)]
        # verification only (not part of the product path): shards concatenate to the full result
        parts = [None] * world
        dist.all_gather_object(parts, (lo, hi, mine))
        q.put((rank, emax, parts, every))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shards_and_max_time():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = torch.arange(24, dtype=torch.float32).reshape(6, 4) * 2.0
    for rank, emax, parts, every in results:
        assert abs(emax - 0.020) < 1e-12                    # MAX over ranks, seen by every rank
        assert every == pytest.approx([0.010, 0.020])       # every rank's own time, in rank order
        cat = torch.cat([m for _, _, m in sorted(parts, key=lambda t: t[0])])
        assert torch.equal(cat, full)
    assert aggregate_throughput(100.0, 10, 0.020, 2) == pytest.approx(100000.0)
    assert gather_over_ranks(0.5) == [0.5]                  # no process group: the one rank's value
