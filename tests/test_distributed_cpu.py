"""CPU, world_size 2, gloo: the multi-GPU path's only inter-rank steps — the
broadcast of the frame-range table from rank 0 and the max/sum reductions of the
timing counters (bev_amd/shard.py, used verbatim by bench.py with backend nccl)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bev_amd import shard


def test_frame_ranges_cover_everything_once():
    for total in [0, 1, 7, 8, 1000, 8000, 8001]:
        for world in [1, 2, 3, 8]:
            t = shard.frame_ranges(total, world)
            assert t.shape == (world, 2) and t[:, 1].sum() == total
            assert (t[1:, 0] == t[:-1, 0] + t[:-1, 1]).all() and t[0, 0] == 0
            assert t[:, 1].max() - t[:, 1].min() <= 1
    assert shard.frame_ranges(8000, 8)[:, 1].tolist() == [1000] * 8   # BASELINE config 4


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = shard.broadcast_ranges(2001, rank, world)        # only rank 0 computes it
        first, count = int(table[rank, 0]), int(table[rank, 1])
        # every rank "processes" its shard: here a checksum of the frame ids it owns
        local = float(np.arange(first, first + count).sum())
        total = shard.sum_over_ranks(local, world)
        slowest = shard.max_over_ranks(1.0 + rank, world)
        rows = shard.gather_per_rank([rank, first, count, 10.0 * rank + 0.5], world)   # bench.py's per-rank report
        assert rows.tolist() == [[0.0, 0.0, 1001.0, 0.5], [1.0, 1001.0, 1000.0, 10.5]]
        q.put((rank, table.tolist(), first, count, total, slowest))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_reductions_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, f0, c0, tot0, s0), (r1, t1, f1, c1, tot1, s1) = res
    assert t0 == t1 == [[0, 1001], [1001, 1000]]          # both ranks hold rank 0's table
    assert (f0, c0, f1, c1) == (0, 1001, 1001, 1000)
    assert tot0 == tot1 == float(np.arange(2001).sum())   # shards are disjoint and complete
    assert s0 == s1 == 2.0                                # max over ranks
