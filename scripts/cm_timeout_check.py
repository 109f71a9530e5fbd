#!/usr/bin/env python3
"""Developer check of the firing-order walk's bounded wait (advisor, round 5): a library built with -DBEV_EXP_NO_REPORTS
(make -C point-cloud-preprocessing-tools_amd exp EXPNAME=noreports EXPFLAGS=-DBEV_EXP_NO_REPORTS) never publishes the
strips' no-return reports, so strip 0 of every frame whose strips talk waits for them in vain: it must give up after
kCmSpins polls — ONCE per frame, not once per band —, fail the frame, and the frame must come out of the general path
equal to the oracle.  Prints the modes and the time per call.
   BEV_AMD_LIB=.../csrc/libbev_noreports.so python3 scripts/cm_timeout_check.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
import bev_amd
import oracle_lib as orc
from bev_amd import synth

p = bev_amd.params_for_sensor("OS1_64")
sp = orc.sensor_from_params(p)
frames = [synth.firing_real(p, 900 + i, noret=0.03) for i in range(6)] + [synth.firing_real(p, 950, noret=0.0)]
ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=p.slots)
for rep in range(2):
    t = time.perf_counter()
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    dt = time.perf_counter() - t
    modes = [int(m) for m in ctx.frame_info(0, len(frames))[:, 1]]
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), i
        assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
    print(f"call {rep}: modes {modes} (2 = gave up and redone, 5 = read in place: no no-return record among its samples), {dt * 1e3:.1f} ms, every frame equal to the oracle")
    assert modes[:6] == [2] * 6 and modes[6] == 5, modes
    assert dt < 2.0, "one bounded wait per frame, not one per band"
ctx.close()
print("ok")
