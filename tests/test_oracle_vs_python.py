"""CPU: the C oracle against an independent plain-Python restatement of the reference text (tests/py_restatement.py)
on small sensors, where the corner cases are dense (duplicates, out-of-range rows / columns, -1 fallbacks with the
C++ `%` sign rule, boundary and non-finite coordinates).  Two restatements written separately from the same source
agreeing bit for bit is not the reference itself (the oracle stays "parity unpinned"), but it removes transcription
slips of either one."""
import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
import py_restatement as py

SPECIAL = np.array([0.0, -0.0, 0.5, -0.5, 1.0, 2.0, -2.0, -75.0, -75.000008, 75.0, -50.0, 50.0, 49.999996, -112.0, -113.0,
                    -112.99999, 111.0, 111.99999, 112.0, 0.29999998, 0.3, 0.30000001, -1.73, 3.75, 3.8750002, 61.75,
                    1e9, -1e9, 3e38, np.inf, -np.inf, np.nan, 1e-40, -1e-40], np.float32)


def _frame(rng, n, h, structured, p_invalid=0.3):
    cnt = int(rng.integers(0, 3 * n * h // 2 + 1))
    pts = np.zeros(cnt, bev_amd.POINT_DTYPE)
    if cnt == 0:
        return pts
    for f in ("x", "y", "z"):
        smooth = rng.normal(0, 20 if f != "z" else 1.5, cnt).astype(np.float32)
        pts[f] = np.where(rng.random(cnt) < 0.3, SPECIAL[rng.integers(0, len(SPECIAL), cnt)], smooth)
    rest = (1.0 - p_invalid) / 3
    pts["intensity"] = rng.choice(np.array([-1.0, 0.0, 0.5, 1.0], np.float32), cnt, p=[p_invalid, rest, rest, rest])
    pts["row"] = rng.integers(0, n + 2, cnt)
    pts["col"] = rng.integers(0, h + 2, cnt)
    pts["t"] = rng.integers(0, 2**32, cnt, dtype=np.uint64).astype(np.uint32)
    pts["label"] = rng.choice(np.array([-2, 0, 1, 7], np.int16), cnt)
    if structured:  # ground-like rings so that phase B / C have something to average
        r = pts["row"].astype(np.float32)
        a = pts["col"].astype(np.float32) * np.float32(2 * np.pi / h)
        rad = np.float32(3) + r * np.float32(2.5)
        keep = rng.random(cnt) < 0.7
        pts["x"] = np.where(keep, rad * np.cos(a), pts["x"]).astype(np.float32)
        pts["y"] = np.where(keep, rad * np.sin(a), pts["y"]).astype(np.float32)
        pts["z"] = np.where(keep, np.float32(-1.7) + rng.normal(0, 0.05, cnt).astype(np.float32), pts["z"])
    return pts


@pytest.mark.parametrize("seed", range(60))
def test_oracle_equals_python_restatement(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 11))
    h = int(rng.integers(5, 48)) if seed % 3 else int(rng.integers(5, 9))   # narrow images: columns 0, 1 and the wrap matter
    g = int(rng.integers(1, n - 1)) if seed % 4 else n - 2                  # every row below row 1 tested
    res = float(rng.choice([0.25, 0.5, 1.0]))
    pts = _frame(rng, n, h, structured=bool(seed % 2), p_invalid=(0.3, 0.6, 0.45)[seed % 3])
    sp = orc.OracleSensor(h, n, g, res)
    o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
    _, _, o_avg = orc.mark_ground(sp, orc.order_cloud(sp, pts))
    p_ord, p_gm, p_avg, p_multi, p_single = py.process_frame(n, h, g, res, pts)
    assert o_ord.tobytes() == p_ord.tobytes(), "ordered cloud / labels"
    assert np.array_equal(o_gm, p_gm), "ground_mat"
    assert np.array_equal(np.asarray(o_avg).reshape(75, 50), p_avg, equal_nan=True), "cell averages"
    assert np.array_equal(o_multi, p_multi), "multi BEV"
    assert np.array_equal(o_single, p_single), "single BEV"


def test_scalar_helpers_match_oracle():
    import ctypes as C
    lib = orc.lib()
    for x in SPECIAL:
        for y in SPECIAL[:12]:
            if not (np.isfinite(x) and np.isfinite(y)):
                continue
            r, c = C.c_int(0), C.c_int(0)
            lib.oracle_belonging_grid(float(x), float(y), C.byref(r), C.byref(c))
            assert (r.value, c.value) == py.belonging_grid(np.float32(x), np.float32(y)), (x, y)
