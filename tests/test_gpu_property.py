"""GPU: property-based parity on tiny sensors.  A range image of a few rows and a few dozen
columns makes the reference's corner cases dense instead of rare: duplicate (row, col) pairs (last writer wins),
out-of-range rows / columns (dropped), intensity == -1 fallbacks including the col < 2 flat-index rule and the
(col + 2) % H wrap, points exactly on cell / layer / image boundaries, non-finite coordinates, empty frames.  Every
example goes through the whole hot path on the device and must equal the oracle bit for bit."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import bev_amd
import oracle_lib as orc

pytestmark = pytest.mark.gpu

# coordinates on and next to the boundaries of the 2 m ground grid (x + 75, y + 50), the 1 m BEV bins (x + 112),
# the 0.3 m height test and the layer / height steps, plus values that overflow the integer conversions
SPECIAL = [0.0, -0.0, 0.5, -0.5, 1.0, 2.0, -2.0, -75.0, -75.000008, 75.0, -50.0, 50.0, 49.999996, -112.0, -113.0, -112.99999, 111.0,
           111.99999, 112.0, 0.29999998, 0.3, 0.30000001, -1.73, -2.0, 3.75, 3.8750002, 61.75, 1e9, -1e9, 3e38,
           float("inf"), float("-inf"), float("nan"), 1e-40, -1e-40]
coord = st.one_of(st.sampled_from(SPECIAL), st.floats(-130, 130, width=32), st.floats(-3, 6, width=32))


@st.composite
def sensor_and_frames(draw):
    n = draw(st.integers(3, 12))
    h = draw(st.integers(5, 70))
    g = draw(st.integers(1, n - 2))
    res = draw(st.sampled_from([0.25, 0.5, 1.0]))
    frames = []
    for _ in range(draw(st.integers(1, 3))):
        cnt = draw(st.integers(0, 3 * n * h // 2))
        seed = draw(st.integers(0, 2**32 - 1))
        rng = np.random.default_rng(seed)
        pts = np.zeros(cnt, bev_amd.POINT_DTYPE)
        if cnt:
            pool = np.array(draw(st.lists(coord, min_size=8, max_size=24)), np.float32)
            for f in ("x", "y", "z"):
                smooth = rng.normal(0, 20 if f != "z" else 1.5, cnt).astype(np.float32)
                pick = rng.integers(0, len(pool), cnt)
                pts[f] = np.where(rng.random(cnt) < 0.35, pool[pick], smooth)
            pts["intensity"] = rng.choice(np.array([-1.0, 0.0, 0.5, 1.0], np.float32), cnt, p=[0.3, 0.1, 0.3, 0.3])
            pts["row"] = rng.integers(0, n + 2, cnt)          # n, n + 1: out of range
            pts["col"] = rng.integers(0, h + 2, cnt)
            pts["t"] = rng.integers(0, 2**32, cnt, dtype=np.uint64).astype(np.uint32)
            pts["label"] = rng.choice(np.array([-2, 0, 1, 7], np.int16), cnt)
            if draw(st.booleans()):                           # ground-like structure: rows at rising radius, flat z
                r = pts["row"].astype(np.float32)
                a = pts["col"].astype(np.float32) * np.float32(2 * np.pi / h)
                rad = np.float32(3) + r * np.float32(2.5)
                keep = rng.random(cnt) < 0.7
                pts["x"] = np.where(keep, rad * np.cos(a), pts["x"]).astype(np.float32)
                pts["y"] = np.where(keep, rad * np.sin(a), pts["y"]).astype(np.float32)
                pts["z"] = np.where(keep, np.float32(-1.7) + rng.normal(0, 0.05, cnt).astype(np.float32), pts["z"])
        frames.append(pts)
    return (n, h, g, res), frames


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(sensor_and_frames())
def test_tiny_sensors_match_oracle(case):
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
            assert np.array_equal(multi[i], o_multi), (n, h, g, i, "multi BEV")
            assert np.array_equal(single[i], o_single), (n, h, g, i, "single BEV")
    finally:
        ctx.close()


# ---- round 4: the same pool of awkward values in the two layouts that are read in place without a sorted prefix:
# structured clouds (KittiPointCloudSelect.cpp:206-207,240) and firing order (MulranPointCloudSelect.cpp:112-130), with and
# without a defect that the probe's samples may or may not see.  Whatever route a frame ends on (3 / 4 read in place, 2 caught
# and redone, 0 not recognised), its outputs equal the oracle's.
import collections                                           # noqa: E402
import os                                                    # noqa: E402

_ROUTES = collections.Counter()


@st.composite
def layout_frames(draw):
    n = draw(st.integers(3, 40))
    h = draw(st.one_of(st.sampled_from([5, 17, 59, 64, 236, 237, 300, 473, 505]),   # one wave, one strip, strip edges on / next to the row end
                       st.integers(5, 1000)))
    g = draw(st.integers(1, n - 2))
    res = draw(st.sampled_from([0.25, 0.5, 1.0]))
    S = n * h
    frames = []
    for _ in range(draw(st.integers(1, 3))):
        rng = np.random.default_rng(draw(st.integers(0, 2**32 - 1)))
        pool = np.array(draw(st.lists(coord, min_size=8, max_size=24)), np.float32)
        pts = np.zeros(S, bev_amd.POINT_DTYPE)
        kind = draw(st.sampled_from(["structured", "firing"]))
        slot = np.arange(S)
        if kind == "structured":
            row, col = slot // h, slot % h
        else:
            row, fire = slot % n, slot // n
            col = fire + rng.integers(0, draw(st.sampled_from([1, 2, 9])), S)     # up to 8 columns past the firing; the last ones out of range
        a = col.astype(np.float32) * np.float32(2 * np.pi / h)
        rad = np.float32(3) + row.astype(np.float32) * np.float32(60.0 / n)
        pts["x"], pts["y"] = rad * np.cos(a), rad * np.sin(a)
        pts["z"] = np.float32(-1.7) + rng.normal(0, 0.06, S).astype(np.float32) + (rng.random(S) < 0.1) * np.float32(1.2)
        for f in ("x", "y", "z"):
            pts[f] = np.where(rng.random(S) < 0.05, pool[rng.integers(0, len(pool), S)], pts[f])
        pts["intensity"] = rng.choice(np.array([-1.0, 0.0, 0.5, 1.0], np.float32), S, p=[0.2, 0.1, 0.35, 0.35])
        pts["row"], pts["col"] = row, col
        pts["t"] = rng.integers(0, 2**32, S, dtype=np.uint64).astype(np.uint32)
        pts["label"] = rng.choice(np.array([-2, -1, 1, 7], np.int16), S, p=[0.7, 0.1, 0.1, 0.1])
        if kind == "structured":
            pts[rng.random(S) < draw(st.sampled_from([0.0, 0.02, 0.5]))] = np.zeros(1, bev_amd.POINT_DTYPE)[0]   # dropped returns
        defect = draw(st.sampled_from(["none", "none", "none", "col", "row", "zero_with_contents", "swap"]))
        if defect != "none" and S > 10:
            i = int(rng.integers(1, S - 1))
            if defect == "col":
                pts["col"][i] = (int(pts["col"][i]) + int(rng.integers(1, 12))) % (h + 2)
            elif defect == "row":
                pts["row"][i] = (int(pts["row"][i]) + 1) % (n + 1)
            elif defect == "zero_with_contents":
                pts["row"][i], pts["col"][i] = 0, 0
            else:
                pts[[i, i + 1]] = pts[[i + 1, i]]
        frames.append(pts)
    return (n, h, g, res, draw(st.booleans())), frames


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(layout_frames())
def test_structured_and_firing_order_layouts_match_oracle(case):
    (n, h, g, res, tile), frames = case   # (tile: a drawn flag that once chose round 4's tile-shaped walk; kept so that the derandomised examples stay the same)
    p = bev_amd.params_for_sensor("HDL_32E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan, p.height_res = n, h, g, res
    sp = orc.sensor_from_params(p)
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(8, n * h))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        for m in ctx.frame_info(0, len(frames))[:, 1]:
            _ROUTES[int(m)] += 1
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes(), (n, h, g, tile, i, "ordered cloud / labels")
            assert np.array_equal(gm[i], o_gm), (n, h, g, tile, i, "ground_mat")
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (n, h, g, tile, i, "BEVs")
    finally:
        ctx.close()


def test_the_layout_examples_took_every_route():
    """(runs after the test above) structured and firing order read in place, defects caught and redone"""
    assert _ROUTES[3] >= 20 and _ROUTES[4] >= 20 and _ROUTES[2] >= 5, dict(_ROUTES)
