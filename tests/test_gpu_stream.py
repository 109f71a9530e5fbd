"""GPU: frames whose points are already in slot order are read in place (k_probe / stream walk) instead of being
scattered through the winner table (getOrderedCloud, BatchMultiBevGen.cpp:102-116).  The path is a GUESS that is
verified on the device; these tests check (a) that sorted sweeps really take it, (b) that inputs which only look sorted
at the sampled positions are caught and redone the general way, and (c) that the results equal the oracle either way."""
import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _stream_on(monkeypatch):
    """reading in place is the default (BEV_STREAM=0, read by bev_create, turns it off): pinned here so that a stray
    environment cannot make these tests pass on the general path alone"""
    monkeypatch.setenv("BEV_STREAM", "1")


def _run(p, ctx, frames):
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    info = ctx.frame_info(0, min(len(frames), max(1, ctx.max_batch // 2)) if len(frames) <= ctx.max_batch // 2 else 1)
    sp = orc.sensor_from_params(p)
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud / labels differ"
        assert np.array_equal(gm[i], o_gm) and np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
    return info


@pytest.mark.parametrize("sensor", ["HDL_64E", "HDL_32E", "OS1_64"])
def test_sorted_sweeps_are_read_in_place(sensor):
    p = bev_amd.params_for_sensor(sensor)
    # (appended points per frame in proportion to the sensor: a (row, strip) lists at most 64 of them, bev_internal.h kTailCap)
    dup = 5000 * p.slots // 133312
    frames = [synth.sweep(p, 70 + i, keep=k, n_dup=d) for i, (k, d) in enumerate([(0.98, dup), (1.0, 0), (0.6, 300), (0.9, 0)])]
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(len(f) for f in frames))
    try:
        info = _run(p, ctx, frames)
        for i, fr in enumerate(frames):
            T, mode, consumed, failed = (int(v) for v in info[i])
            # (the full sweep without appended points is exactly S records, every one its slot's point: a structured
            # cloud, tests/test_gpu_structured.py)
            assert mode == (3 if i == 1 else 1) and failed == 0 and consumed == T, (i, info[i])
            assert len(fr) - T <= 5000  # everything but the appended duplicates (k_probe follows the prefix to its exact end)
    finally:
        ctx.close()


def test_labels_other_than_minus_two_survive_the_in_place_path():
    """phase C puts a candidate's own label back when it un-grounds it; in place there is no winner table to find the
    input point through, so the walk must not have overwritten a label it cannot reconstruct (anything but -2)"""
    p = bev_amd.params_for_sensor("HDL_64E")
    rng = np.random.default_rng(5)
    frames = []
    for i in range(3):
        f = synth.sweep(p, 200 + i, keep=0.95, n_dup=2000)
        f["label"] = rng.choice(np.array([-2, -1, 0, 1, 7], dtype=np.int16), size=len(f), p=[0.5, 0.2, 0.1, 0.1, 0.1])
        frames.append(f)
    # (repeated: a register copy made while its asm load was still in flight once gave wrong tail points in one run of
    # six — a timing-dependent fault that a single pass does not show)
    for _ in range(12):
        ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(len(f) for f in frames))
        try:
            info = _run(p, ctx, frames)
            assert [int(m) for m in info[:3, 1]] == [1, 1, 1]
        finally:
            ctx.close()


def test_unsorted_frames_go_the_general_way():
    p = bev_amd.params_for_sensor("OS1_64")
    frames = [synth.firing_order(p, 3), synth.adversarial(p, 60000, 1, False), synth.sweep(p, 4)[:1500]]
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=70000)
    try:
        info = _run(p, ctx, frames)
        assert [int(m) for m in info[:, 1]] == [4, 0, 0]   # (firing order has a route of its own: tests/test_gpu_colmajor.py)
    finally:
        ctx.close()


def test_inputs_that_only_look_sorted_are_caught_and_redone():
    """The probe samples every 63rd point; everything in between is verified by the walk.  Each frame below is a
    sorted sweep with one defect hidden from the samples."""
    p = bev_amd.params_for_sensor("HDL_64E")
    base = synth.sweep(p, 90, keep=0.97, n_dup=0)
    H = p.horizon_scan

    def swapped(i, j):
        f = base.copy()
        f[[i, j]] = f[[j, i]]
        return f

    def dup_slot(i):  # two consecutive prefix points in the same slot: the later one must win
        f = base.copy()
        f[i + 1]["row"], f[i + 1]["col"] = f[i]["row"], f[i]["col"]
        return f

    def oob(i):
        f = base.copy()
        f[i]["row"] = 200
        return f

    def shifted_block(i, n, d):  # n points claim columns d further right: a gap and an overlap inside the prefix
        f = base.copy()
        f["col"][i:i + n] = np.minimum(f["col"][i:i + n] + d, H - 1)
        return f

    frames = [swapped(1001, 1002), swapped(50001, 50300), dup_slot(70001), oob(33333), shifted_block(90001, 40, 7),
              swapped(3, 4), base]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=len(base))
    try:
        info = _run(p, ctx, frames)
        modes = [int(m) for m in info[:, 1]]
        assert modes[-1] == 1                       # the clean sweep is read in place
        assert all(m == 2 for m in modes[:-1]), modes  # every defect is caught, the frame redone
    finally:
        ctx.close()


def test_gaps_that_shift_a_window_off_its_halo_columns_are_caught_or_harmless():
    """A (row, strip)'s window is placed by interpolation between the probe's samples; the walk counts and order-checks the
    points of the strip's OWN columns only.  A run of dropped returns next to a row start or a strip boundary moves the
    estimate by tens of positions while every own point may still lie inside the 256-position window — and the two halo
    columns on either side (whose points feed phase A's (c + 2) % H and c - 2 fallbacks, BatchMultiBevGen.cpp:146-154)
    may not.  The window has to bracket its whole span or the frame is redone; either way the result is the oracle's.
    The points next to the gaps carry intensity -1, so that the fallbacks are really taken."""
    p = bev_amd.params_for_sensor("HDL_64E")
    H, strip = p.horizon_scan, 236
    for fid, keep in [(300, 1.0), (301, 0.98), (302, 0.9)]:
        frames = []
        base = synth.sweep(p, fid, keep=keep, n_dup=0)
        slot = base["row"].astype(np.int64) * H + base["col"]
        for variant in range(4):
            drop = np.zeros(len(base), bool)
            minus1 = np.zeros(len(base), bool)
            for r in range(15, 64, 3):
                if variant == 0:    # 100 empty slots just after the row's start
                    lo, hi = r * H + 2, r * H + 102
                elif variant == 1:  # ... just before a strip boundary (strip 3 | strip 4)
                    lo, hi = r * H + 4 * strip - 104, r * H + 4 * strip - 4
                elif variant == 2:  # ... just before the row's end: the next row's strip 0 reads its flat-index halo there
                    lo, hi = r * H + H - 110, r * H + H - 3
                else:               # ... right after a strip boundary, 40 wide (estimate too high for the strip's left halo)
                    lo, hi = r * H + 5 * strip + 1, r * H + 5 * strip + 41
                drop |= (slot >= lo) & (slot < hi)
                # upper neighbours (row r - 1) of the row's first and last columns and of the strip's edge columns: invalid
                for c in (0, 1, H - 2, H - 1, 4 * strip - 2, 4 * strip - 1, 4 * strip, 4 * strip + 1, 5 * strip, 5 * strip + 1):
                    minus1 |= (slot == (r - 1) * H + c) | (slot == r * H + c)
            f = base.copy()
            f["intensity"][minus1] = -1.0
            frames.append(np.ascontiguousarray(f[~drop]))
        ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))  # one sub-batch
        try:
            info = _run(p, ctx, frames)   # every frame equal to the oracle
            modes = [int(m) for m in info[:, 1]]
            assert all(m in (1, 2) for m in modes), modes       # sorted sweeps: read in place or caught and redone
        finally:
            ctx.close()


def test_a_last_strip_of_one_column_leaves_the_wrap_around_halo_to_the_strip_before():
    """H mod 236 == 1: column H - 2 belongs to the strip BEFORE the last, and its (c + 2) % H fallback
    (BatchMultiBevGen.cpp:146-149) is column 0 — that strip's virtual column H.  Found by the round-4 property test on a
    473-column sensor (until then only the last strip fetched a wrap-around halo in the indexed sources).  Invalid returns
    at (r, H - 2) make the fallback count; sorted sweeps with a tail and firing order, both read in place."""
    for n, h, g in [(24, 473, 4), (40, 709, 30), (16, 237, 10)]:
        p = bev_amd.params_for_sensor("HDL_32E")
        p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
        rng = np.random.default_rng(h)
        frames = []
        for fid, dup in [(1, 0), (2, 300), (3, 0)]:
            f = synth.sweep(p, 400 + fid, keep=1.0 if fid != 3 else 0.9, n_dup=dup)
            edge = (f["col"] >= h - 3) | (f["col"] <= 2)
            f["intensity"][edge & (rng.random(len(f)) < 0.7)] = -1.0
            frames.append(f)
        fo = synth.firing_order(p, 404)
        edge = (fo["col"] >= h - 3) | (fo["col"] <= 2)
        fo["intensity"][edge & (rng.random(len(fo)) < 0.7)] = -1.0
        frames.append(fo)
        ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
        try:
            info = _run(p, ctx, frames)
            assert [int(m) for m in info[:, 1]] == [3, 1, 1, 4], (n, h, g, info)   # (the full sweep without a tail is a structured cloud)
        finally:
            ctx.close()


def test_stream_and_general_frames_mixed_in_one_sub_batch_and_the_knob():
    import os
    p = bev_amd.params_for_sensor("HDL_32E")
    # (9000 appended points over 32 rows x 5 strips: more than a (row, strip) can list -> that frame goes the general way)
    frames = [synth.sweep(p, 1, n_dup=1200), synth.firing_order(p, 2), synth.sweep(p, 3, keep=0.5, n_dup=1500), np.empty(0, bev_amd.POINT_DTYPE),
              synth.sweep(p, 5, n_dup=1200), synth.sweep(p, 6, keep=0.5, n_dup=9000)]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
    try:
        info = _run(p, ctx, frames)
        assert [int(m) for m in info[:, 1]] == [1, 4, 1, 0, 1, 0]
        assert int(info[5, 2]) == 4  # the reason k_probe gives: a tail list overflowed
    finally:
        ctx.close()
    os.environ["BEV_STREAM"] = "0"
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
    try:
        info = _run(p, ctx, frames)
        assert [int(m) for m in info[:, 1]] == [0, 0, 0, 0, 0, 0]
    finally:
        ctx.close()


# ---- property-based: sorted clouds of mid-sized sensors with everything the tiny-sensor test (test_gpu_property.py) throws
# at the general path — boundary coordinates, non-finite values, intensity -1, mixed labels — plus appended out-of-order
# points (also out of range) and hidden defects.  Whatever mode a frame ends in, its outputs must equal the oracle.
from hypothesis import HealthCheck, given, settings          # noqa: E402
from hypothesis import strategies as st                      # noqa: E402

import collections                                           # noqa: E402
_MODES = collections.Counter()
_SPECIAL = np.array([0.0, -0.0, 0.5, -75.0, -75.000008, 75.0, -50.0, 49.999996, -112.0, -112.99999, 111.99999, 112.0,
                     0.29999998, 0.3, -1.73, -2.0, 3.8750002, 1e9, -3e38, np.inf, -np.inf, np.nan, 1e-40], np.float32)


@st.composite
def _sorted_frames(draw):
    n = draw(st.integers(8, 40))
    h = draw(st.one_of(st.sampled_from([300, 473, 504, 505, 506, 709, 757, 1024]),   # 2 .. 5 strips, strip edges on / next to the row end (473, 709: the last strip owns ONE column)
                       st.integers(230, 1100)))
    g = draw(st.integers(1, n - 2))
    frames = []
    for _ in range(draw(st.integers(1, 3))):
        rng = np.random.default_rng(draw(st.integers(0, 2**32 - 1)))
        keep = draw(st.sampled_from([1.0, 0.97, 0.8, 0.5]))
        slots = np.nonzero(rng.random(n * h) < keep)[0]
        n_tail = draw(st.sampled_from([0, 0, 40, 900, 3000]))
        cnt = len(slots) + n_tail
        pts = np.zeros(cnt, bev_amd.POINT_DTYPE)
        row = np.concatenate([slots // h, rng.integers(0, n + 2, n_tail)])      # tail: anywhere, also out of range
        col = np.concatenate([slots % h, rng.integers(0, h + 2, n_tail)])
        a = col.astype(np.float32) * np.float32(2 * np.pi / h)
        rad = np.float32(3) + row.astype(np.float32) * np.float32(80.0 / n)
        pts["x"], pts["y"] = rad * np.cos(a), rad * np.sin(a)
        pts["z"] = np.float32(-1.7) + rng.normal(0, 0.08, cnt).astype(np.float32) + (rng.random(cnt) < 0.1) * np.float32(1.5)
        for f in ("x", "y", "z"):
            odd = rng.random(cnt) < 0.03
            pts[f] = np.where(odd, _SPECIAL[rng.integers(0, len(_SPECIAL), cnt)], pts[f])
        pts["intensity"] = rng.choice(np.array([-1.0, 0.0, 0.5, 1.0], np.float32), cnt, p=[0.15, 0.05, 0.4, 0.4])
        pts["row"], pts["col"] = row, col
        pts["t"] = rng.integers(0, 2**32, cnt, dtype=np.uint64).astype(np.uint32)
        pts["label"] = rng.choice(np.array([-2, -1, 0, 1, 7], np.int16), cnt, p=[0.6, 0.1, 0.1, 0.1, 0.1])
        defect = draw(st.sampled_from(["none", "none", "swap", "dup", "oob"]))
        if len(slots) > 3000 and defect != "none":
            i = int(rng.integers(1500, len(slots) - 1000))
            if defect == "swap":
                pts[[i, i + 1]] = pts[[i + 1, i]]
            elif defect == "dup":
                pts["row"][i + 1], pts["col"][i + 1] = pts["row"][i], pts["col"][i]
            else:
                pts["row"][i] = n + 5
        frames.append(pts)
    return (n, h, g), frames


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(_sorted_frames())
def test_sorted_clouds_of_mid_sized_sensors_match_oracle(case):
    (n, h, g), frames = case
    p = bev_amd.params_for_sensor("HDL_32E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
    sp = orc.sensor_from_params(p)
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(8, max(len(f) for f in frames)))  # one chunk, one sub-batch
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        for m in ctx.frame_info(0, len(frames))[:, 1]:
            _MODES[int(m)] += 1
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes(), (n, h, g, i, "ordered cloud / labels")
            assert np.array_equal(gm[i], o_gm), (n, h, g, i, "ground_mat")
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (n, h, g, i, "BEVs")
    finally:
        ctx.close()


def test_the_property_examples_took_all_three_routes():
    """(runs after the test above: read in place, general, caught and redone)"""
    assert _MODES[1] >= 10 and _MODES[0] >= 1 and _MODES[2] >= 1, dict(_MODES)
