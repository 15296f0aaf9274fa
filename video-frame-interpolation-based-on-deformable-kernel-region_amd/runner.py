"""Multi-GPU runner: frame pairs are independent units, sharded over ranks with no
data-path collective ("replicas only"; the reference is a single
process on a single device -- demo_MiddleBury.py:254 loops over pairs one by one).

One process per GPU.  The only communication is the timing protocol of the bench: a barrier on
both sides of the timed region, a MAX reduction of the per-rank wall time and a SUM of the units
processed -- three host-side scalars, sent over `gloo` (TCP on the loopback): the data path has no
exchange step, so RCCL is never initialised and xGMI stays idle.

Ranks come either from an external launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment) or from `spawn_ranks`, which starts them itself -- before anything in the parent
has touched a GPU -- and joins them.
"""
import os
import socket
import subprocess
import sys
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
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_externally():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, world_size, extra_env=None, timeout=None):
    """Start `world_size` copies of `python argv...`, one per GPU (RANK = LOCAL_RANK = i, a free
    rendezvous port on 127.0.0.1), wait for all of them and return the largest exit code.  Rank 0's
    stdout is the caller's stdout.  A rank that fails takes the others down with it.  The caller must
    not have initialised a GPU (children inherit nothing but the environment)."""
    port = free_port()
    procs = []
    for r in range(world_size):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world_size),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = None if timeout is None else time.time() + timeout
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = max(worst, rc if rc > 0 else 1)
                for q in live:                       # the exact children started above, nothing else
                    q.terminate()
        if deadline is not None and time.time() > deadline:
            for q in live:
                q.kill()
            return max(worst, 124)
        time.sleep(0.02)
    return worst


def init_distributed(backend="gloo"):
    """Joins the process group when the launcher made more than one rank.  gloo by default: the
    bench's join is a barrier and two scalar reductions on host memory."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: start the ranks with torchrun or runner.spawn_ranks")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def timed_region(step_fn, steps, device=None):
    """barrier + sync, `steps` calls of step_fn(i), sync + barrier; returns the MAX wall time
    over ranks in seconds (every rank gets the same number)."""
    cuda = device is not None and device.type == "cuda"
    if cuda:
        torch.cuda.synchronize(device)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    if cuda:
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def total_units(local_units):
    """SUM over ranks of the units each processed (host-side join, not a data-path collective)."""
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([float(local_units)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
    return float(local_units)


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
