#!/usr/bin/env python3
"""bev_process_batch through one long-lived context per sensor: random batch sizes (chunks, sub-batches and the staging's two
halves all come into play), random subsets of the outputs requested, every layout mixed, the single-cloud entry points in
between — every output against the oracle.  usage (GPU box): python3 scripts/api_fuzz.py [seed] [calls] [sensor]"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd")); sys.path.insert(0, str(REPO / "tests"))
import numpy as np
import bev_amd, oracle_lib as orc
from bev_amd import synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sensor = sys.argv[3] if len(sys.argv) > 3 else "HDL_32E"
rng = np.random.default_rng(seed)
p = bev_amd.params_for_sensor(sensor)
sp = orc.sensor_from_params(p)
S = p.slots
max_batch = int(rng.choice([2, 7, 16, 40]))
ctx = bev_amd.BevContext(p, device=0, max_batch=max_batch, max_points=2 * S)
bad = 0
t0 = time.time()


def make(fid):
    k = int(rng.integers(0, 7))
    if k == 0: return synth.sweep(p, fid, keep=float(rng.choice([1.0, 0.98, 0.6])), n_dup=int(rng.choice([0, 40, 400])))
    if k == 1: return synth.structured(p, fid, float(rng.choice([1.0, 0.9])), kitti_intensity=bool(fid % 2))
    if k == 2: return synth.firing_order(p, fid)
    if k == 3: return synth.adversarial(p, int(rng.integers(1, 2 * S)), fid, bool(fid % 2))
    if k == 4: return np.empty(0, bev_amd.POINT_DTYPE)
    if k == 5: return synth.sweep(p, fid)[: int(rng.integers(1, 3000))]
    return synth.concat(p, fid, n_sweeps=2, keep=0.9)[: 2 * S]


for call in range(calls):
    n = int(rng.integers(1, 3 * max_batch + 2))
    frames = [make(1000 * call + i) for i in range(n)]
    wm, ws, wg = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ordered, multi, single, gm = ctx.process_batch(frames, want_multi=wm, want_single=ws, want_ground_mat=wg)
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        ok = ordered[i].tobytes() == o_ord.tobytes()
        ok = ok and (not wm or np.array_equal(multi[i], o_multi)) and (not ws or np.array_equal(single[i], o_single)) and (not wg or np.array_equal(gm[i], o_gm))
        if not ok:
            bad += 1
            print(f"MISMATCH call {call} frame {i} of {n} (max_batch {max_batch}, want {wm, ws, wg}, {len(pts)} points)", flush=True)
    if call % 5 == 0:   # the single-cloud entry points on the same context
        pts = frames[int(rng.integers(0, n))]
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        got = ctx.order_cloud(pts)
        plain = orc.order_cloud(sp, pts)
        lab, g2 = ctx.mark_ground(plain)
        ok = got.tobytes() == plain.tobytes() and lab.tobytes() == o_ord.tobytes() and np.array_equal(g2, o_gm)
        ok = ok and np.array_equal(ctx.multi_bev(o_ord), o_multi) and np.array_equal(ctx.single_bev(o_ord), o_single)
        if not ok:
            bad += 1
            print(f"MISMATCH single-cloud entry points after call {call}", flush=True)
ctx.close()
print(f"api fuzz ({sensor}, max_batch {max_batch}): seed {seed}, {calls} calls, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
