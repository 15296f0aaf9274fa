"""Multi-GPU runner: frame pairs are independent units, sharded over ranks with no
data-path collective ("replicas only"; the reference is a single
process on a single device -- demo_MiddleBury.py:254 loops over pairs one by one).

One process per GPU (torch.distributed: backend "nccl" = RCCL on the GPU box, "gloo" in
the CPU tests).  The only communication is the timing protocol of the bench: a barrier on
both sides of the timed region and a MAX reduction of the per-rank wall time.
"""
import os
import time

import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world_size):
    """Contiguous, balanced slice of range(n_pairs) owned by `rank` (sizes differ by <= 1)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(n_pairs, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def timed_region(step_fn, steps, device=None):
    """barrier + sync, `steps` calls of step_fn(i), sync + barrier; returns the MAX wall time
    over ranks in seconds (every rank gets the same number)."""
    cuda = device is not None and device.type == "cuda"
    barrier(device)
    if cuda:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    if cuda:
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier(device)
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def total_units(local_units, device=None):
    """SUM over ranks of the units each processed (host-side join, not a data-path collective)."""
    if dist.is_available() and dist.is_initialized():
        cuda = device is not None and device.type == "cuda"
        t = torch.tensor([float(local_units)], dtype=torch.float64, device=device if cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
    return float(local_units)
