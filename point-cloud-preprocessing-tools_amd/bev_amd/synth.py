"""ctypes binding of synth/libbev_synth.so: deterministic synthetic frames
(BASELINE.json configs).  Test / bench input generation only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import POINT_DTYPE, SYNTH_PATH, BevError, BevParams

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not SYNTH_PATH.exists():
            raise BevError(f"{SYNTH_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(str(SYNTH_PATH))
        P = C.POINTER(BevParams)
        lib.bev_synth_sweep.argtypes = [P, C.c_uint64, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.bev_synth_sweep.restype = C.c_size_t
        lib.bev_synth_firing_order.argtypes = [P, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.bev_synth_firing_order.restype = C.c_size_t
        lib.bev_synth_firing_real.argtypes = [P, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_size_t]
        lib.bev_synth_firing_real.restype = C.c_size_t
        lib.bev_synth_concat.argtypes = [P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p, C.c_size_t]
        lib.bev_synth_concat.restype = C.c_size_t
        lib.bev_synth_adversarial.argtypes = [P, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
        lib.bev_synth_adversarial.restype = C.c_size_t
        _lib = lib
    return _lib


SEED_BASE = 0xBEEF0000  # seed = 0xBEEF0000 + frame_idx


def sweep(params: BevParams, frame_id: int, keep: float = 0.98, n_dup: int = 5000, seed: int = SEED_BASE,
          out: np.ndarray | None = None) -> np.ndarray:
    """Structured sweep (configs 1, 2, 4): row-major kept slots + appended duplicates."""
    lib = _load()
    cap = params.slots + n_dup
    buf = out if out is not None else np.empty(cap, dtype=POINT_DTYPE)
    n = lib.bev_synth_sweep(C.byref(params), seed, frame_id, keep, n_dup, buf.ctypes.data, buf.shape[0])
    return buf[:n]


def sweep_unique(params: BevParams, frame_id: int, n_points: int, seed: int = SEED_BASE) -> np.ndarray:
    """Config 1: exactly n_points real points with unique (row, col)."""
    keep = min(1.0, 1.25 * n_points / params.slots)
    pts = sweep(params, frame_id, keep=keep, n_dup=0, seed=seed)
    if len(pts) < n_points:
        pts = sweep(params, frame_id, keep=1.0, n_dup=0, seed=seed)
    # deterministic thinning: keep every k-th until n_points remain
    idx = (np.arange(n_points, dtype=np.int64) * len(pts)) // n_points
    return np.ascontiguousarray(pts[idx])


def firing_order(params: BevParams, frame_id: int, seed: int = SEED_BASE) -> np.ndarray:
    """MulRan-style unordered cloud (config 3)."""
    lib = _load()
    buf = np.empty(params.slots, dtype=POINT_DTYPE)
    n = lib.bev_synth_firing_order(C.byref(params), seed, frame_id, buf.ctypes.data, buf.shape[0])
    return buf[:n]


def firing_real(params: BevParams, frame_id: int, noret: float = 0.03, phase: int | None = None, direction: int | None = None,
                stagger: float = 1.0, seed: int = SEED_BASE) -> np.ndarray:
    """What mulran_point_cloud_select writes for a real Ouster sweep (MulranPointCloudSelect.cpp:112-130): firing order with
    an arbitrary start azimuth (`phase`, columns; default: drawn from the frame id), either direction of rotation (default:
    drawn), the four staggered laser columns (+9, +3, -3, -9 columns by beam mod 4, scaled by `stagger`) and a share `noret`
    of no-return records (x = y = z = 0 -> atan2(0, 0) = 0 -> column 0 of their row)."""
    lib = _load()
    h = (frame_id * 2654435761 + 12345) & 0xFFFFFFFF
    if phase is None:
        phase = h % params.horizon_scan
    if direction is None:
        direction = 1 if (h >> 20) & 1 else -1
    buf = np.empty(params.slots, dtype=POINT_DTYPE)
    n = lib.bev_synth_firing_real(C.byref(params), seed, frame_id, noret, int(phase), int(direction), stagger, buf.ctypes.data, buf.shape[0])
    return buf[:n]


def concat(params: BevParams, frame_id: int, n_sweeps: int = 60, keep: float = 0.98, seed: int = SEED_BASE) -> np.ndarray:
    """Oxford-style concatenated sweeps (config 5): P >> S."""
    lib = _load()
    cap = params.slots * n_sweeps
    buf = np.empty(cap, dtype=POINT_DTYPE)
    n = lib.bev_synth_concat(C.byref(params), seed, frame_id, n_sweeps, keep, buf.ctypes.data, buf.shape[0])
    return buf[:n]


def adversarial(params: BevParams, n_points: int, seed: int, nonfinite: bool = False) -> np.ndarray:
    """Edge-case cloud: OOB rows/cols, duplicates, labels incl. 0, boundary coordinates."""
    lib = _load()
    buf = np.empty(n_points, dtype=POINT_DTYPE)
    n = lib.bev_synth_adversarial(C.byref(params), seed, n_points, 1 if nonfinite else 0, buf.ctypes.data, n_points)
    return buf[:n]


def structured(params: BevParams, frame_id: int, keep: float = 0.98, kitti_intensity: bool = False,
               seed: int = SEED_BASE) -> np.ndarray:
    """A structured cloud of exactly S records, what the KITTI selector writes (KittiPointCloudSelect.cpp:206-207,240):
    record i is the point of slot i or — a dropped return — an all-zero record (row = col = 0).  The kept points are those
    of `sweep` with the same frame id.  kitti_intensity: every real point carries intensity -1 (:238)."""
    pts = sweep(params, frame_id, keep=keep, n_dup=0, seed=seed)
    out = np.zeros(params.slots, dtype=POINT_DTYPE)
    out[pts["row"].astype(np.int64) * params.horizon_scan + pts["col"]] = pts
    if kitti_intensity:
        out["intensity"][out["label"] == -2] = -1.0
    return out
