"""Frame sharding for the multi-GPU path.

Frames are independent (reference main(), BatchMultiBevGen.cpp:727-757, carries no
state between iterations), so the only inter-rank traffic is the broadcast of
the frame-range table from rank 0 and the reduction of timing counters.  With
backend "nccl" these run over RCCL/xGMI; the CPU tests use gloo."""
from __future__ import annotations

import numpy as np


def frame_ranges(total_frames: int, world_size: int) -> np.ndarray:
    """Contiguous, balanced [first, count] per rank over a sorted frame list."""
    base, extra = divmod(total_frames, world_size)
    counts = np.array([base + (1 if r < extra else 0) for r in range(world_size)], dtype=np.int64)
    firsts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    return np.stack([firsts, counts], axis=1)


def broadcast_ranges(total_frames: int, rank: int, world_size: int, device="cpu", force_collective=False) -> np.ndarray:
    """Rank 0 computes the table and broadcasts it (the one collective of the path)."""
    import torch
    import torch.distributed as dist

    table = torch.zeros((world_size, 2), dtype=torch.int64, device=device)
    if rank == 0:
        table.copy_(torch.from_numpy(frame_ranges(total_frames, world_size)))
    if world_size > 1 or force_collective:
        dist.broadcast(table, src=0)
    return table.cpu().numpy()


def max_over_ranks(value: float, world_size: int, device="cpu", force_collective=False) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world_size > 1 or force_collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_rank(values, world_size: int, device="cpu", force_collective=False) -> np.ndarray:
    """Every rank's row of float64 values, on every rank: (world_size, len(values)).  Reporting only (what each rank
    measured: bench.py's line shows that the collective really ran over world_size ranks)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([list(values)], dtype=torch.float64, device=device)
    if world_size > 1 or force_collective:
        out = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(out, t)
        t = torch.cat(out, dim=0)
    return t.cpu().numpy()


def sum_over_ranks(value: float, world_size: int, device="cpu", force_collective=False) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world_size > 1 or force_collective:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
