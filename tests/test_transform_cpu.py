"""CPU: the rigid transform of cloud_manip (CloudManip.cpp:119-128).  The oracle's matrix and per-point arithmetic are
checked against an independent numpy float32 restatement, and the product library's host-side matrix builder
(bev_yaw_translate_matrix, no device needed) against the oracle."""
import numpy as np

import bev_amd
import oracle_lib as orc
from bev_amd import synth

F = np.float32


def _matrix_numpy(tx, ty, tz, yaw):
    """Affine3f = Identity; translation << t; rotate(AngleAxisf(theta, UnitZ())), Eigen 3.3 formulas, float32."""
    theta = F(np.float64(F(yaw) / F(180.0)) * np.pi)           # :124 float / float, double * M_PI, stored to float
    s, c = np.sin(theta, dtype=F), np.cos(theta, dtype=F)      # glibc sinf / cosf through numpy's float32 loops
    axis = np.array([0, 0, 1], F)
    sin_axis, cos1_axis = (s * axis).astype(F), (F(F(1) - c) * axis).astype(F)
    r = np.zeros((3, 3), F)
    tmp = F(cos1_axis[0] * axis[1]); r[0, 1] = F(tmp - sin_axis[2]); r[1, 0] = F(tmp + sin_axis[2])
    tmp = F(cos1_axis[0] * axis[2]); r[0, 2] = F(tmp + sin_axis[1]); r[2, 0] = F(tmp - sin_axis[1])
    tmp = F(cos1_axis[1] * axis[2]); r[1, 2] = F(tmp - sin_axis[0]); r[2, 1] = F(tmp + sin_axis[0])
    for k in range(3):
        r[k, k] = F(F(cos1_axis[k] * axis[k]) + c)
    m = np.zeros((3, 4), F)
    m[:, :3] = r            # Identity * R
    m[:, 3] = [tx, ty, tz]
    return m.reshape(12)


def _transform_numpy(cloud, m):
    out = cloud.copy()
    x, y, z = cloud["x"], cloud["y"], cloud["z"]
    with np.errstate(all="ignore"):
        for k, name in enumerate("xyz"):
            a, b, c, t = (F(v) for v in m[4 * k:4 * k + 4])
            out[name] = (a * x).astype(F) + ((b * y).astype(F) + ((c * z).astype(F) + t).astype(F)).astype(F)
    return out


CASES = [(0, 0, 0, 0), (1.5, -2.25, 0.125, 30), (0, 0, 0, 90), (-3, 4, 1, -45.5), (10, 20, -1, 180), (0.1, 0.2, 0.3, 359.9),
         (0, 0, 0, 1e-3), (5, 5, 5, -720.25)]


def test_oracle_matrix_matches_numpy_restatement():
    for tx, ty, tz, yaw in CASES:
        got, want = orc.yaw_translate_matrix(tx, ty, tz, yaw), _matrix_numpy(tx, ty, tz, yaw)
        assert got.tobytes() == want.tobytes() or np.array_equal(got, want), (yaw, got, want)
    m = orc.yaw_translate_matrix(0, 0, 0, 0)
    assert np.array_equal(m.reshape(3, 4), np.eye(3, 4, dtype=F))
    # the diagonal entry of the axis is (1 - c) + c in float32: not always a literal 1
    assert any(orc.yaw_translate_matrix(0, 0, 0, y)[10] != 1 for y in np.linspace(1, 179, 400)), "m22 never rounds"


def test_library_matrix_equals_oracle():
    rng = np.random.default_rng(3)
    for tx, ty, tz, yaw in CASES + [tuple(rng.uniform(-400, 400, 4)) for _ in range(300)]:
        a, b = bev_amd.yaw_translate_matrix(tx, ty, tz, yaw), orc.yaw_translate_matrix(tx, ty, tz, yaw)
        assert np.array_equal(a, b), (yaw, a, b)


def test_oracle_transform_matches_numpy_restatement():
    p = bev_amd.params_for_sensor("HDL_32E")
    for seed, (tx, ty, tz, yaw) in enumerate(CASES):
        cloud = synth.adversarial(p, 5000, seed, nonfinite=(seed % 2 == 1))
        m = orc.yaw_translate_matrix(tx, ty, tz, yaw)
        got, want = orc.transform_cloud(cloud, m), _transform_numpy(cloud, m)
        assert got.tobytes() == want.tobytes() or all(
            np.array_equal(got[f], want[f], equal_nan=True) for f in ("x", "y", "z")), seed
        for f in ("intensity", "row", "col", "t", "label"):
            assert np.array_equal(got[f], cloud[f], equal_nan=True)
    assert len(orc.transform_cloud(np.empty(0, bev_amd.POINT_DTYPE), m)) == 0
