"""CPU: SURVEY §8(f) row N3 — the selectors' range-image projection.  (1) csrc/bev_libm.h reproduces the
host libm's atanf for all 2^32 floats and atan2f on ~4e8 random / structured / special pairs;
(2) the projection helpers built on it equal the oracle's literal restatement (libm) point for point."""
import ctypes as C

import numpy as np
import pytest

import hostcheck_lib as hc
import oracle_lib as orc
from projection_data import KITTI_VARIANTS, kitti_returns, raw_returns


def test_atanf_and_atan2f_are_bit_identical_to_this_libm():
    out = (C.c_uint64 * 2)()
    hc.lib().hc_libm_vs_host(400_000_000, out)
    assert (int(out[0]), int(out[1])) == (0, 0)


@pytest.mark.parametrize("kind", [0, 1])
def test_projection_equals_oracle(kind):
    for seed in range(3):
        pts = raw_returns(200_000, seed)
        xyzi = pts if kind == 0 else np.ascontiguousarray(pts.T)
        got, want = hc.project(kind, xyzi), orc.project(kind, xyzi)
        assert got.tobytes() == want.tobytes()
        ok = np.isfinite(pts).all(axis=1)
        if kind == 0:
            assert (want["row"] == np.arange(len(pts)) % 64).all()
            assert want["col"][ok].max() == 1024          # the overflow column exists (dropped later by the bounds test)
        else:
            assert want["row"].max() == 31 and want["col"][ok].max() == 1055
            assert (want["x"][ok] == -pts[ok, 0]).all() and (want["z"][ok] == -pts[ok, 2]).all()
        assert (want["label"] == -2).all()


# ---- KITTI: the ring index is a sequential counter of azimuth zero crossings (KittiPointCloudSelect.cpp:186-243) ----
def _kitti_python(xyzi):
    """The reference loop in plain Python (libm's atan2f through ctypes for the azimuth), for small inputs."""
    libm = C.CDLL("libm.so.6")
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    libm.atan2f.restype = C.c_float
    H, N = 2083, 64
    out = {}
    n = len(xyzi)
    az = [np.float32(float(libm.atan2f(float(p[1]), float(p[0]))) / np.pi * 180.0) for p in xyzi]
    ring = 0 if az[0] > 0 else -1
    count = 0
    for i in range(1, n):
        if az[i - 1] <= 0 and az[i] > 0:
            if ring == -1:
                ring, count = 0, 0
            elif np.float32(count) > np.float32(H) * np.float32(0.60):
                ring, count = ring + 1, 0
        a = az[i]
        a = a - np.float32(360) if a >= 360 else (a + np.float32(360) if a < 0 else a)
        col = int(np.floor(abs(float(a) / (360.0 / H)) + 0.5))      # std::round for a non-negative argument
        if 0 <= ring < N:
            col = col - H if col >= H else col
            out[ring * H + col] = i
        count += 1
    return out


@pytest.mark.parametrize("start", ["after_seam", "before_seam"])
def test_kitti_oracle_against_python_restatement(start):
    rng = np.random.default_rng(5)
    chunks = [rng.uniform(200, 359, 30)] if start == "before_seam" else []
    for m in (1400, 900, 1300, 1251, 1249, 1600):   # 900 and 1249 are too short to close a ring, 1251 is just enough
        chunks.append((np.arange(m) + rng.uniform(0.1, 0.9, m)) / m * 360.0)
    a = np.deg2rad(np.concatenate(chunks))
    xyzi = np.stack([10 * np.cos(a), 10 * np.sin(a), rng.normal(0, 1, len(a)), rng.random(len(a))], 1).astype(np.float32)
    want = _kitti_python(xyzi)
    got = orc.project(2, xyzi)
    filled = np.flatnonzero(got["label"] == -2)
    assert sorted(want) == filled.tolist()
    for slot, i in want.items():
        g = got[slot]
        assert (g["x"], g["y"], g["z"]) == tuple(xyzi[i, :3]) and g["intensity"] == -1.0
        assert (g["row"], g["col"]) == divmod(slot, 2083)
    # rings: 1400 | 900 + 1300 (the crossing after 900 points is ignored) | 1251 | 1249 + 1600 -> 4 rings
    assert np.unique(got["row"][filled]).tolist() == [0, 1, 2, 3]
    empty = np.delete(got, filled)
    assert not empty.tobytes().strip(b"\0")


@pytest.mark.parametrize("variant", KITTI_VARIANTS)
def test_kitti_decomposition_equals_oracle(variant):
    """crossing lists -> chain of accepted crossings -> ring by counting links (the kernels' decomposition, run on the
    host) against the sequential loop."""
    for seed in range(3):
        xyzi = kitti_returns(seed, variant)
        assert hc.project(2, xyzi).tobytes() == orc.project(2, xyzi).tobytes()
    xyzi = kitti_returns(7, variant)
    for n in (0, 1, 2, 3, 255, 256, 257, 1249, 1250, 1251, 1252, 2600, 40_000):
        assert hc.project(2, xyzi[:n]).tobytes() == orc.project(2, xyzi[:n]).tobytes(), n


def test_kitti_ring_acceptance_limit():
    # `num_points_on_this_ring > Horizon_SCAN * 0.60f` with an int on the left: true from 1250 on
    out = (C.c_uint32 * 1)()
    hc.lib().hc_kitti_ring_min(out)
    assert out[0] == 1250
