"""GPU: the BEV code lists between the walk / phase C and the rasters (computeAndSave{Multi,Single}Bev,
BatchMultiBevGen.cpp:271-292, 340-356).  A (writer, raster band) list holds kCodeListCap codes (bev_internal.h); a frame
with a list that does not hold its codes is flagged and its images are computed from the ordered, labelled cloud instead.
Results must not depend on which way a frame's codes travelled."""
import os

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu


def _run(p, frames, cap=None):
    old = os.environ.get("BEV_CODE_CAP")
    if cap is not None:
        os.environ["BEV_CODE_CAP"] = str(cap)
    try:
        ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(8, max(len(f) for f in frames)))
    finally:
        if cap is not None:
            if old is None:
                del os.environ["BEV_CODE_CAP"]
            else:
                os.environ["BEV_CODE_CAP"] = old
    try:
        assert len(frames) < 8
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        ovf = ctx.code_overflow(0, len(frames))
        sp = orc.sensor_from_params(p)
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), i
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
        # the single-cloud entry point that runs the identity walk (no k_probe: the host clears the overflow counter)
        o_ord = orc.process_frame(sp, frames[0])[0]
        plain = orc.order_cloud(sp, frames[0])
        got, _ = ctx.mark_ground(plain)
        assert got.tobytes() == o_ord.tobytes()
    finally:
        ctx.close()
    return [int(v) for v in ovf]


@pytest.mark.parametrize("cap", [1, 37, 600])
def test_tiny_lists_flag_every_frame(cap):
    p = bev_amd.params_for_sensor("HDL_64E")
    frames = [synth.sweep(p, 80), synth.structured(p, 81, 0.9), synth.firing_order(p, 82), synth.adversarial(p, 90000, 5, True),
              np.empty(0, bev_amd.POINT_DTYPE), synth.sweep(p, 83, keep=0.3, n_dup=100)]
    ovf = _run(p, frames, cap)
    assert all(v > 0 for v in ovf[:3]) and ovf[4] == 0, ovf   # (600: a benchmark frame's fullest lists are longer)
    if cap == 1:
        assert all(v > 0 for i, v in enumerate(ovf) if i != 4), ovf


def test_benchmark_frames_fit_their_lists_and_a_pile_in_one_band_does_not():
    p = bev_amd.params_for_sensor("HDL_64E")
    H, N = p.horizon_scan, p.n_scan
    pile = synth.structured(p, 84, 1.0)     # every return above ONE fine raster band (7 image rows), all heights and columns
    rng = np.random.default_rng(1)
    pile["x"] = rng.uniform(-10.9, -4.1, len(pile)).astype(np.float32)   # image rows 102 .. 108: fine band 5
    pile["y"] = rng.uniform(-100, 100, len(pile)).astype(np.float32)
    pile["z"] = rng.uniform(-2, 3.9, len(pile)).astype(np.float32)
    ovf = _run(p, [synth.sweep(p, 85), synth.sweep(p, 86, keep=1.0, n_dup=0), synth.structured(p, 87, 0.98), pile])
    assert ovf[:3] == [0, 0, 0], ovf      # the layouts bench.py runs fit their lists
    assert ovf[3] != 0, ovf
    p = bev_amd.params_for_sensor("OS1_64")   # (the sensor whose lists run fullest: a slower, equal raster if one overflows)
    assert _run(p, [synth.firing_order(p, i) for i in range(3)] + [synth.sweep(p, i) for i in range(3)]) == [0] * 6
