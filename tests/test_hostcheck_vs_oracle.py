"""CPU: the closed forms the kernels use (csrc/bev_exact.h, composed in
tests/hostcheck) against the sequential oracle, bit for bit."""
import numpy as np
import pytest

import bev_amd
import hostcheck_lib as hc
import oracle_lib as orc
from bev_amd import synth

SENSORS = ["HDL_32E", "HDL_64E", "OS1_64"]


def _compare(p, pts):
    sp = orc.sensor_from_params(p)
    o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
    _, _, o_avg = orc.mark_ground(sp, orc.order_cloud(sp, pts))
    h_ord, h_gm, h_avg, h_multi, h_single = hc.process_frame(p, pts)
    assert h_avg.tobytes() == o_avg.tobytes(), "per-cell average heights differ"
    assert np.array_equal(h_gm, o_gm), f"ground_mat differs at {np.argwhere(h_gm != o_gm)[:5]}"
    assert h_ord.tobytes() == o_ord.tobytes(), "ordered cloud / labels differ"
    assert np.array_equal(h_multi, o_multi)
    assert np.array_equal(h_single, o_single)
    return o_gm


@pytest.mark.parametrize("sensor", SENSORS)
def test_sweep(sensor):
    p = bev_amd.params_for_sensor(sensor)
    for fid in range(2):
        gm = _compare(p, synth.sweep(p, fid))
        assert (gm == 1).sum() > 1000 and (gm == -1).sum() > 100  # the frame exercises all three states


@pytest.mark.parametrize("sensor", SENSORS)
def test_firing_order(sensor):
    p = bev_amd.params_for_sensor(sensor)
    _compare(p, synth.firing_order(p, 3))


@pytest.mark.parametrize("sensor", SENSORS)
@pytest.mark.parametrize("nonfinite", [False, True])
def test_adversarial(sensor, nonfinite):
    p = bev_amd.params_for_sensor(sensor)
    for seed in range(3):
        _compare(p, synth.adversarial(p, 60000, seed, nonfinite))


def test_concat_small():
    p = bev_amd.params_for_sensor("HDL_32E")
    _compare(p, synth.concat(p, 0, n_sweeps=6))


def test_empty_and_degenerate():
    p = bev_amd.params_for_sensor("HDL_32E")
    _compare(p, np.empty(0, bev_amd.POINT_DTYPE))
    one = synth.sweep(p, 0)[:1]
    _compare(p, one)
    allsame = synth.sweep(p, 0)[:5000].copy()
    allsame["row"] = 31
    allsame["col"] = 0
    _compare(p, allsame)
    noret = synth.sweep(p, 1).copy()
    noret["intensity"] = -1.0  # what the KITTI selector writes on every real point
    _compare(p, noret)
    lab0 = synth.sweep(p, 2).copy()
    lab0["label"] = 0
    _compare(p, lab0)
