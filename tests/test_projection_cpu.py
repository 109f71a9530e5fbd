"""CPU: SURVEY §8(f) row N3 — the selectors' range-image projection.  (1) csrc/bev_libm.h reproduces the
host libm's atanf for all 2^32 floats and atan2f on ~4e8 random / structured / special pairs;
(2) the projection helpers built on it equal the oracle's literal restatement (libm) point for point."""
import ctypes as C

import numpy as np
import pytest

import hostcheck_lib as hc
import oracle_lib as orc
from projection_data import raw_returns


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
