"""CPU: SURVEY §8(f) row N1 — the label step (host/LabelStep.cpp) against the reference's OWN k-NN.

oracle/_ref/libref_knn.so is built (oracle/Makefile, `make ref`) from the reference's include/nanoflann.hpp and
include/KDTreeVectorOfVectorsAdaptor.h where they lie under /root/reference, and called like
BatchMultiBevGen.cpp:534-550 / :594-613 call it.  The loops around the searches (selectMajorFrames :502-566,
getKeyFrameLabel :575-636) are restated below in Python, float32 like the reference; the product implementation
(exhaustive scan in nanoflann's distance evaluation order) must give the same major frames and the same labels,
bit for bit.  The .so is prebuilt where /root/reference is absent; without it the tests are skipped."""
import ctypes as C
import pathlib
import subprocess

import numpy as np
import pytest

import hostcheck_lib as hc

ROOT = pathlib.Path(__file__).resolve().parents[1]
REF_SO = ROOT / "oracle" / "_ref" / "libref_knn.so"


@pytest.fixture(scope="module")
def ref_knn():
    if pathlib.Path("/root/reference/include/nanoflann.hpp").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], check=True, capture_output=True)
    if not REF_SO.exists():
        pytest.skip("oracle/_ref/libref_knn.so not built (needs /root/reference)")
    lib = C.CDLL(str(REF_SO))
    lib.ref_knn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.ref_knn.restype = C.c_int

    def knn(pts, q, k):
        pts = np.ascontiguousarray(pts, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros(k, np.uint64)
        d2 = np.zeros(k, np.float32)
        found = lib.ref_knn(pts.ctypes.data, len(pts), q.ctypes.data, k, idx.ctypes.data, d2.ctypes.data)
        return idx.astype(np.int64), d2, found

    return knn


def _reference_loops(xyz, knn):
    """BatchMultiBevGen.cpp:502-566 and :575-636 around the real k-NN."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    f32 = np.float32
    major = [0]
    for i in range(1, len(xyz)):
        df = (xyz[i] - xyz[major[-1]]).astype(np.float32)          # getDistance, src/Utility.cpp:43-49
        dist = f32(np.sqrt(f32(f32(f32(df[0] * df[0]) + f32(df[1] * df[1])) + f32(df[2] * df[2]))))
        if dist < f32(20):                                          # :527-531
            continue
        _, d2, _ = knn(xyz[major], xyz[i], 1)                       # :534-550
        if d2[0] < f32(400):
            continue
        major.append(i)
    labels = np.zeros((len(xyz), len(major)), np.float32)
    for i in range(len(xyz)):
        idx, d2, _ = knn(xyz[major], xyz[i], 2)                     # :604-613
        if i == major[idx[0]]:
            labels[i, idx[0]] = 1.0                                 # :616-618
        else:
            w0 = f32(1.0 / (np.float64(d2[0]) + 1e-5))              # :623-627
            w1 = f32(1.0 / (np.float64(d2[1]) + 1e-5))
            s = f32(w0 + w1)
            labels[i, idx[0]] = f32(w0 / s)
            labels[i, idx[1]] = f32(w1 / s)
    return np.array(major, np.int32), labels


def _trajectory(seed, n, revisit=True):
    rng = np.random.default_rng(seed)
    step = rng.normal(0, 1, (n, 3)) * [3.0, 3.0, 0.2] + [2.5, 0.5, 0.0]
    xyz = np.cumsum(step, axis=0)
    if revisit:                                                     # drive back over the first part of the route
        back = xyz[: n // 3][::-1] + rng.normal(0, 0.7, (n // 3, 3))
        xyz = np.concatenate([xyz, back])
    return xyz.astype(np.float32)


@pytest.mark.parametrize("seed", range(4))
def test_label_step_equals_reference_knn_loops(ref_knn, seed):
    xyz = _trajectory(seed, 600)
    want_major, want_labels = _reference_loops(xyz, ref_knn)
    got_major = hc.select_major_frames(xyz)
    assert got_major.tolist() == want_major.tolist()
    assert 5 < len(want_major) < len(xyz)                           # revisits were rejected, the route was long enough
    got_labels = hc.keyframe_labels(xyz, got_major)
    assert got_labels.tobytes() == want_labels.tobytes()
    assert np.allclose(got_labels.sum(axis=1), 1.0, atol=1e-6)


def test_label_step_short_inputs(ref_knn):
    for n in (1, 2, 3):
        xyz = _trajectory(9, 40, revisit=False)[:: 40 // n][:n]
        want_major, want_labels = _reference_loops(xyz, ref_knn)
        got_major = hc.select_major_frames(xyz)
        assert got_major.tolist() == want_major.tolist()
        assert hc.keyframe_labels(xyz, got_major).tobytes() == want_labels.tobytes()


def test_distances_are_nanoflanns(ref_knn):
    """The exhaustive scan must reproduce nanoflann's squared distances bit for bit (its L2 adaptor adds the three
    squared differences in index order, in float)."""
    rng = np.random.default_rng(3)
    pts = (rng.normal(0, 200, (500, 3))).astype(np.float32)
    for q in (rng.normal(0, 200, (50, 3))).astype(np.float32):
        idx, d2, found = ref_knn(pts, q, 2)
        assert found == 2
        diff = (q - pts[idx]).astype(np.float32)
        mine = np.float32(np.float32(diff[:, 0] * diff[:, 0]) + np.float32(diff[:, 1] * diff[:, 1])) + np.float32(diff[:, 2] * diff[:, 2])
        assert mine.astype(np.float32).tobytes() == d2.tobytes()
        full = ((pts - q) ** 2).sum(axis=1)
        assert set(idx.tolist()) == set(np.argsort(full)[:2].tolist())


def test_exact_ties_follow_the_kd_tree(ref_knn):
    """Positions on a 10 m lattice: many major frames are EXACTLY equidistant from a key frame, so which one gets the
    weight is decided by nanoflann's tree (partition order inside the leaves, traversal order), not by the distance.
    More than ten major frames, so the tree has inner nodes."""
    rng = np.random.default_rng(0)
    for trial in range(6):
        steps = rng.integers(-1, 2, (400, 3)) * [10, 10, 0]
        xyz = np.cumsum(steps, axis=0).astype(np.float32)
        want_major, want_labels = _reference_loops(xyz, ref_knn)
        assert len(want_major) > 12
        got_major = hc.select_major_frames(xyz)
        assert got_major.tolist() == want_major.tolist()
        assert hc.keyframe_labels(xyz, got_major).tobytes() == want_labels.tobytes()
    # and the raw search: every query, k = 1 and 2, indices included
    pts = (rng.integers(-6, 7, (300, 3)) * 10).astype(np.float32)
    lib = hc.lib()
    lib.hc_knn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    for q in (rng.integers(-7, 8, (200, 3)) * 5).astype(np.float32):
        for k in (1, 2, 5):
            idx, d2, _ = ref_knn(pts, q, k)
            gi, gd = np.zeros(k, np.uint64), np.zeros(k, np.float32)
            lib.hc_knn(pts.ctypes.data, len(pts), q.ctypes.data, k, gi.ctypes.data, gd.ctypes.data)
            assert gi.astype(np.int64).tolist() == idx.tolist() and gd.tobytes() == d2.tobytes(), (q, k)
