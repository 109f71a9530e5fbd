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


def test_candidate_key_reproduces_the_bev_code():
    """csrc/bev_exact.h: a candidate travels as height + key; the resolve kernel rebuilds its BEV code from them.
    Round trip on points on and around every cell / bin boundary, clamped cells, labels 0 / -2 / others, non-finite
    coordinates, three raster intervals; on real frames nearly every candidate takes the fast decode, the rest escape
    to reading the point."""
    rng = np.random.default_rng(5)
    for interval in (1.0, 0.5, 2.0):
        p = bev_amd.params_for_sensor("HDL_64E")
        p.interval = interval
        xs = np.concatenate([np.arange(-130, 131, 0.5), np.nextafter(np.arange(-130, 131, 1.0, dtype=np.float32), np.float32(1e9)),
                             np.nextafter(np.arange(-130, 131, 1.0, dtype=np.float32), np.float32(-1e9)),
                             rng.uniform(-120, 120, 4000), [np.inf, -np.inf, np.nan, 1e30, -1e30, 0.0, -0.0, 1e-40]]).astype(np.float32)
        ys = rng.permutation(xs)
        zs = rng.uniform(-3, 6, len(xs)).astype(np.float32)
        res = np.array([hc.key_roundtrip(p, x, y, z, lab) for x, y, z in zip(xs, ys, zs) for lab in (-2, 0, 7)])
        assert (res != 0).all(), (interval, np.flatnonzero(res == 0)[:5])
        if interval == 1.0:
            assert (res == 2).mean() < 0.6  # escapes are the clamped cells / out-of-grid points of this artificial set
    # real frames: fast decodes dominate
    p = bev_amd.params_for_sensor("HDL_64E")
    hc.key_stats(reset=True)
    for fid in range(3):
        _compare(p, synth.sweep(p, 100 + fid, keep=0.98, n_dup=5000))
    dec, esc = hc.key_stats()
    assert dec > 1000 and esc < 0.25 * (dec + esc), (dec, esc)
    print(f"candidate keys on 3 HDL_64E frames: {dec} codes rebuilt from key + height, {esc} escapes")
