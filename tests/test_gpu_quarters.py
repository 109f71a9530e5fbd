"""GPU: phase B (markGroundPoints' per-cell float sums, BatchMultiBevGen.cpp:187-210) cut into four small workgroups per
frame — cells with cell mod 4 == q belong to workgroup q, which reads run q of every candidate segment (BEV_CS_QUARTERS=1,
read by bev_create).  Cells are independent and the order inside a cell is the slot order either way, so every output
must equal the oracle exactly as with the one-workgroup form."""
import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _quarters_on(monkeypatch):
    monkeypatch.setenv("BEV_CS_QUARTERS", "1")


@pytest.mark.parametrize("sensor", ["HDL_64E", "HDL_32E", "OS1_64"])
def test_quarter_workgroups_equal_the_oracle(sensor):
    p = bev_amd.params_for_sensor(sensor)
    S = p.n_scan * p.horizon_scan
    frames = [synth.sweep(p, 31, keep=0.98, n_dup=3000), synth.firing_order(p, 32), synth.adversarial(p, S // 2, 33, True),
              synth.sweep(p, 34, keep=0.4, n_dup=0), np.empty(0, bev_amd.POINT_DTYPE), synth.sweep(p, 36, keep=1.0, n_dup=0)]
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(len(f) for f in frames))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        sp = orc.sensor_from_params(p)
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud / labels differ"
            assert np.array_equal(gm[i], o_gm) and np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
    finally:
        ctx.close()


# the tiny-sensor property examples of test_gpu_property.py (duplicates, out-of-range points, intensity -1 fall-backs,
# boundary and non-finite coordinates, empty frames) through the four-workgroup form
from hypothesis import HealthCheck, given, settings     # noqa: E402
from test_gpu_property import sensor_and_frames          # noqa: E402


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(sensor_and_frames())
def test_tiny_sensors_match_oracle_with_quarter_workgroups(case):
    (n, h, g, res), frames = case
    p = bev_amd.params_for_sensor("HDL_32E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan, p.height_res = n, h, g, res
    sp = orc.sensor_from_params(p)
    ctx = bev_amd.BevContext(p, device=0, max_batch=2, max_points=max(8, max(len(f) for f in frames)))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes(), (n, h, g, i, "ordered cloud / labels")
            assert np.array_equal(gm[i], o_gm), (n, h, g, i, "ground_mat")
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (n, h, g, i, "BEVs")
    finally:
        ctx.close()
