#!/usr/bin/env python3
"""Random SENSOR GEOMETRIES (N_SCAN, Horizon_SCAN, GROUND_UPPER_SCAN, HEIGHT_RES, interval) through the whole hot path, every
layout, every frame against the oracle: what depends on how the columns fall into strips, waves and raster bands (round 4's
one-column last strip was such a case).  Column counts are biased towards multiples of 236 (the walk's strip), 59 (a quarter strip:
wave) and 64, plus or minus a few.
usage (GPU box): python3 scripts/geometry_soak.py [seed] [geometries]"""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd")); sys.path.insert(0, str(REPO / "tests"))
import numpy as np
import bev_amd, oracle_lib as orc
from bev_amd import synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
modes = {}
for it in range(count):
    while True:
        n = int(rng.choice([rng.integers(3, 17), rng.integers(3, 65), rng.integers(65, 129)]))
        kind = int(rng.integers(0, 4))
        if kind == 0:   h = int(236 * rng.integers(1, 9) + rng.integers(-3, 4))
        elif kind == 1: h = int(59 * rng.integers(1, 30) + rng.integers(-2, 3))
        elif kind == 2: h = int(64 * rng.integers(1, 33) + rng.integers(-2, 3))
        else:           h = int(rng.integers(5, 3000))
        h = max(5, h)
        g = int(rng.integers(1, n - 1))
        strips = (h + 235) // 236
        if n * h <= (1 << 18) and (g + 1) * strips <= 1024:
            break
    p = bev_amd.params_for_sensor("HDL_32E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
    p.height_res = float(rng.choice([0.25, 0.5, 1.0]))
    p.interval = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
    sp = orc.sensor_from_params(p)
    inv = lambda f: (f.__setitem__("intensity", np.where(rng.random(len(f)) < 0.15, np.float32(-1), f["intensity"])), f)[1]
    st = synth.structured(p, it, float(rng.choice([1.0, 0.97, 0.6])))
    real = st["label"] == -2
    st["intensity"][real & (rng.random(len(st)) < 0.15)] = -1.0
    frames = [inv(synth.sweep(p, it, keep=float(rng.choice([1.0, 0.98, 0.7])), n_dup=int(rng.choice([0, 50, 700])))), st,
              inv(synth.firing_order(p, it)), synth.adversarial(p, int(rng.integers(1, 2 * n * h + 2)), it, bool(it % 2)),
              np.empty(0, bev_amd.POINT_DTYPE), inv(synth.sweep(p, it + 7, keep=0.9, n_dup=0))]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(8, max(len(f) for f in frames)))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        info = ctx.frame_info(0, len(frames))
        M = p.mat_size
        for i, pts in enumerate(frames):
            modes[int(info[i, 1])] = modes.get(int(info[i, 1]), 0) + 1
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            ok = ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm)
            if p.interval == 1.0:
                ok = ok and np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single)
            else:   # (the oracle's per-frame entry point rasters at 1.0: the other intervals through its raster functions)
                ok = ok and np.array_equal(multi[i], orc.multi_bev(sp, o_ord, p.interval).reshape(p.n_layers, M, M)) \
                        and np.array_equal(single[i], orc.single_bev(o_ord, p.interval).reshape(M, M))
            if not ok:
                bad += 1
                print(f"MISMATCH geometry N={n} H={h} G={g} res={p.height_res} interval={p.interval} frame {i} mode {int(info[i, 1])}", flush=True)
    finally:
        ctx.close()
    if (it + 1) % 50 == 0:
        print(f"  ... {it + 1} geometries, {bad} mismatches, modes {modes}, {time.time() - t0:.0f} s", flush=True)
print(f"geometry soak: seed {seed}, {count} geometries x 6 frames, {bad} mismatches, routes taken {modes}")
sys.exit(1 if bad else 0)
