#!/usr/bin/env python3
"""How the frames of the benchmark workload take the in-place path: mode histogram (0 general, 1 in place, 2 caught and
redone) of n HDL_64E sweeps (keep 0.98, 5000 appended points) and, for frames that did not stay in place, why."""
import collections
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd"))
sys.path.insert(0, str(REPO / "tests"))
os.environ.setdefault("BEV_STREAM", "1")
import numpy as np  # noqa: E402
import bev_amd  # noqa: E402
from bev_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bev_amd.params_for_sensor("HDL_64E")
frames = [synth.sweep(p, i, keep=0.98, n_dup=5000) for i in range(n)]
ctx = bev_amd.BevContext(p, device=0, max_batch=2 * n, max_points=max(len(f) for f in frames))
ctx.process_batch(frames)
n = n // 2  # (frame_info describes the last sub-batch: the second half of the one chunk)
info = ctx.frame_info(0, n)
frames = frames[n:]
modes = collections.Counter(int(m) for m in info[:, 1])
print("modes", dict(modes))
for i in range(n):
    T, mode, consumed, failed = (int(v) for v in info[i])
    if mode != 1:
        print(i, "T", T, "n", len(frames[i]), "mode", mode, "consumed", consumed, "failed", failed)
ctx.close()
