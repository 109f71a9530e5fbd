"""CPU: the oracle's two stated assumptions about the phase-A angle test (BatchMultiBevGen.cpp:173,179), MEASURED.

1. Overload reading.  BatchMultiBevGen.h:38 has `using namespace std` commented out, so the unqualified sqrt / atan2 /
   abs bind to the float overloads or to the C double functions depending on which third-party header pulled in
   libstdc++'s <math.h>.  The oracle adopts the float reading; `oracle_angle_is_ground_f64` /
   `oracle_mark_ground_variant(..., ORACLE_ANGLE_F64)` is the other one.  This test counts the slots on which the two
   differ over all 1000 BASELINE frames, the adversarial set and a +-8 ulp sweep around the 10 degree cut.
   MEASURED (glibc 2.35, this image): 3 of the 133,312,000 slots of the 1000 BASELINE frames get a different ground_mat
   value and a different label (frames printed by the test); everything else is identical.  So the choice of reading is
   observable at the 2e-8 level — not zero — and stays an assumption of the oracle ("Oracle").
2. libm.  Both readings go through the libm of the machine the test runs on; its version is part of every failure
   message (the device predicate's constant was derived on glibc 2.35, tests/test_angle_predicate.py).

This pins nothing to the reference binary (nothing here can); it turns an assumption into a number."""
import platform
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import bev_amd
import oracle_lib as orc
from bev_amd import synth

LIBC = "libc: %s %s" % platform.libc_ver()


def _both(sp, pts):
    ordered = orc.order_cloud(sp, pts)
    a, gma, _ = orc.mark_ground(sp, ordered, 0)
    b, gmb, _ = orc.mark_ground(sp, ordered, 1)
    return a, gma, b, gmb


def test_baseline_frames_overload_readings_differ_in_at_most_ten_slots():
    p = bev_amd.params_for_sensor("HDL_64E")
    sp = orc.sensor_from_params(p)

    def one(f):
        a, gma, b, gmb = _both(sp, synth.sweep(p, f, keep=0.98, n_dup=5000))
        return int((gma != gmb).sum()), int((a["label"] != b["label"]).sum()), f

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, range(1000)))
    gm_diff = sum(r[0] for r in res)
    label_diff = sum(r[1] for r in res)
    print(f"overload readings over 1000 BASELINE HDL_64E frames ({LIBC}): {gm_diff} ground_mat slots differ, "
          f"{label_diff} labels differ, of {1000 * p.slots} slots")
    print("frames with a differing label:", [r[2] for r in res if r[1]])
    # slots differ only where an input sits within 1 ulp of the cut: a handful per 133 M slots (3 with glibc 2.35);
    # a label can only follow a flipped slot
    assert gm_diff <= 10 and label_diff <= gm_diff, (gm_diff, label_diff, LIBC)


def test_other_workloads_and_adversarial_set():
    total = 0
    for sensor, make in (("OS1_64", lambda p, f: synth.firing_order(p, f)),
                         ("HDL_32E", lambda p, f: synth.sweep(p, f)),
                         ("HDL_64E", lambda p, f: synth.adversarial(p, 60000 + 911 * f, f, f % 2 == 1))):
        p = bev_amd.params_for_sensor(sensor)
        sp = orc.sensor_from_params(p)
        for f in range(40):
            a, gma, b, gmb = _both(sp, make(p, f))
            n = int((gma != gmb).sum())
            total += n
            # adversarial clouds place many points on exact boundaries; labels may follow a flipped slot there,
            # so only the count is bounded
            assert n <= 4, (sensor, f, n, LIBC)
    print(f"overload readings, other workloads ({LIBC}): {total} ground_mat slots differ")


def test_threshold_sweep_readings_differ_only_within_one_ulp_of_the_cut():
    lib = orc.lib()
    rng = np.random.default_rng(7)
    n = 300_000
    s = np.ldexp(1.0 + rng.random(n), rng.integers(-12, 12, n)).astype(np.float32)
    cut = np.float64(np.tan(np.deg2rad(10.0)))
    t = (cut * s.astype(np.float64)).astype(np.float32)
    dz = (t.view(np.uint32).astype(np.int64) + rng.integers(-8, 9, n)).astype(np.uint32).view(np.float32)
    dz = np.where(rng.random(n) < 0.5, -dz, dz).astype(np.float32)
    ang = rng.random(n) * 2 * np.pi
    dx = (s.astype(np.float64) * np.cos(ang)).astype(np.float32)
    dy = (s.astype(np.float64) * np.sin(ang)).astype(np.float32)
    f32 = np.fromiter((lib.oracle_angle_is_ground(float(a), float(b), float(c)) for a, b, c in zip(dx, dy, dz)), np.uint8, n)
    f64 = np.fromiter((lib.oracle_angle_is_ground_f64(float(a), float(b), float(c)) for a, b, c in zip(dx, dy, dz)), np.uint8, n)
    differ = np.flatnonzero(f32 != f64)
    print(f"+-8 ulp sweep ({LIBC}): {len(differ)} of {n} inputs judged differently by the two readings")
    # where they differ, the exact ratio |dz| / sqrt(dx^2 + dy^2) is within a few float ulps of tan(10 deg)
    horiz = np.sqrt(dx[differ].astype(np.float64) ** 2 + dy[differ].astype(np.float64) ** 2)
    rel = np.abs(np.abs(dz[differ].astype(np.float64)) / horiz / cut - 1.0)
    assert (rel < 4 * 2.0 ** -23).all(), (rel.max() if len(rel) else 0, LIBC)
    assert len(differ) < n // 10, (len(differ), LIBC)
    # far from the cut they always agree
    far = np.abs(np.abs(dz.astype(np.float64)) / np.sqrt(dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2) / cut - 1.0) > 1e-5
    assert np.array_equal(f32[far], f64[far])
