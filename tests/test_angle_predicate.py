"""CPU: the transcendental-free phase-A angle test (csrc/bev_exact.h) against the
reference expression evaluated with THIS machine's libm (reference
BatchMultiBevGen.cpp:169-179).  If a different libm ever changes a last-ulp
result of atanf/atan2f near 10 degrees, these tests say so."""
import ctypes as C

import numpy as np

import hostcheck_lib as hc
import oracle_lib as orc


def test_threshold_constants_match_this_libm():
    out = (C.c_uint64 * 5)()
    hc.lib().hc_derive_angle_threshold(out)
    ta, qs, bad_q, nonmono, bad_a = [int(v) for v in out]
    assert bad_a == 0, "the degrees conversion + compare is not a prefix of the non-negative floats"
    assert nonmono == 0, "atanf is not monotone on this libm: the ratio test is not equivalent"
    assert bad_q == 0
    assert ta == 0x3E32B8C2, hex(ta)                      # largest accepted angle, 0x1.657184p-3 rad
    assert qs == hc.lib().hc_tan_threshold_bits(), hex(qs)  # == kTanThresholdBits compiled into the kernels


def test_ratio_test_equals_atan2f_expression():
    assert hc.lib().hc_angle_vs_libm(400_000_000, 12345) == 0


def test_oracle_function_agrees_on_a_sample():
    rng = np.random.default_rng(7)
    n = 100000
    s = np.ldexp(1.0 + rng.random(n), rng.integers(-10, 10, n)).astype(np.float32)
    t = (np.float64(0.17632698070846498) * s.astype(np.float64)).astype(np.float32)
    dz = (t.view(np.uint32) + rng.integers(-4, 5, n)).astype(np.uint32).view(np.float32)
    dx, dy = s, np.zeros(n, np.float32)
    mine = hc.angle(dx, dy, dz)
    lib = orc.lib()
    ref = np.fromiter((lib.oracle_angle_is_ground(float(a), 0.0, float(c)) for a, c in zip(dx, dz)), np.uint8, n)
    assert np.array_equal(mine, ref)
    assert 0.2 < ref.mean() < 0.8  # the sample really straddles the threshold


def test_special_values():
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 3e38, 1e-38], np.float32)
    g = np.array(np.meshgrid(sp, sp, sp)).reshape(3, -1)
    lib = orc.lib()
    ref = np.fromiter((lib.oracle_angle_is_ground(float(a), float(b), float(c)) for a, b, c in zip(*g)), np.uint8,
                      g.shape[1])
    assert np.array_equal(hc.angle(g[0], g[1], g[2]), ref)
