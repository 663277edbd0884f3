"""Multi-GPU driver: polynomials are independent, so the batch dimension is sharded in
contiguous row blocks, one process per GPU, with NO collective on the compute path
(SURVEY.md §8e).  torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU)
is used only for the optional scatter of a, b from a root / gather of c, and for the
max-over-ranks timing reduction in bench.py.
"""
from __future__ import annotations

import os
from typing import Tuple


def shard_rows(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first_row, row_count) of this rank's contiguous block; block sizes differ by at most one."""
    base, extra = divmod(int(batch), int(world_size))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str = None):
    """One process per GPU; rendezvous from MASTER_ADDR/MASTER_PORT (torchrun sets them)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def scatter_rows(full, batch: int, n: int, dtype, device, src: int = 0):
    """Optional: root holds the full [batch, n] tensor; every rank receives its row block.
    Point-to-point sends from the root (xGMI is point-to-point: one link per peer)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    start, count = shard_rows(batch, world, rank)
    mine = torch.empty((count, n), dtype=dtype, device=device)
    if rank == src:
        reqs = []
        for r in range(world):
            s, c = shard_rows(batch, world, r)
            if r == src:
                mine.copy_(full[s:s + c])
            elif c:
                reqs.append(dist.isend(full[s:s + c].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif count:
        dist.recv(mine, src=src)
    return mine


def gather_rows(mine, batch: int, n: int, dst: int = 0):
    """Optional: collect every rank's output block on the root as one [batch, n] tensor."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        full = torch.empty((batch, n), dtype=mine.dtype, device=mine.device)
        for r in range(world):
            s, c = shard_rows(batch, world, r)
            if r == dst:
                full[s:s + c].copy_(mine)
            elif c:
                buf = torch.empty((c, n), dtype=mine.dtype, device=mine.device)
                dist.recv(buf, src=r)
                full[s:s + c].copy_(buf)
        return full
    if mine.shape[0]:
        dist.send(mine.contiguous(), dst=dst)
    return None


def max_over_ranks(value: float, device=None) -> float:
    """bench.py contract: the job time is the slowest rank's time."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(values, device=None):
    """Every rank's list of floats, in rank order, on every rank (bench.py: per-rank rows and kernel times)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(map(float, values))]
    t = torch.tensor(list(map(float, values)), dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]
