"""ctypes loader for oracle/libbev_oracle.so — the CPU checker (tests only)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from bev_amd import POINT_DTYPE

ORACLE_SO = Path(__file__).resolve().parent.parent / "oracle" / "libbev_oracle.so"


class OracleSensor(C.Structure):
    _fields_ = [("horizon_scan", C.c_int), ("n_scan", C.c_int), ("ground_upper_scan", C.c_int),
                ("height_res", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(str(ORACLE_SO))
        SP = C.POINTER(OracleSensor)
        vp, sz = C.c_void_p, C.c_size_t
        l.oracle_sensor_params.argtypes = [C.c_int, SP]
        l.oracle_belonging_grid.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.oracle_belonging_grid.restype = None
        l.oracle_order_cloud.argtypes = [SP, vp, sz, vp]
        l.oracle_order_cloud.restype = None
        l.oracle_angle_is_ground.argtypes = [C.c_float, C.c_float, C.c_float]
        l.oracle_mark_ground.argtypes = [SP, vp, vp, vp]
        l.oracle_mark_ground.restype = None
        l.oracle_mark_ground_variant.argtypes = [SP, vp, vp, vp, C.c_int]
        l.oracle_mark_ground_variant.restype = None
        l.oracle_angle_is_ground_f64.argtypes = [C.c_float, C.c_float, C.c_float]
        l.oracle_save_bin_csv.argtypes = [vp, vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        l.oracle_multi_bev.argtypes = [SP, vp, sz, C.c_float, vp]
        l.oracle_multi_bev.restype = None
        l.oracle_single_bev.argtypes = [vp, sz, C.c_float, vp]
        l.oracle_single_bev.restype = None
        l.oracle_process_frame.argtypes = [SP, vp, sz, vp, vp, vp, vp]
        l.oracle_process_frame.restype = None
        l.oracle_float_bev.argtypes = [vp, sz, C.c_float, C.c_int, vp]
        l.oracle_float_bev.restype = None
        l.oracle_yaw_translate_matrix.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, vp]
        l.oracle_yaw_translate_matrix.restype = None
        l.oracle_transform_cloud.argtypes = [vp, sz, vp, vp]
        l.oracle_transform_cloud.restype = None
        l.oracle_project_mulran.argtypes = [vp, sz, vp]
        l.oracle_project_mulran.restype = None
        l.oracle_project_oxford.argtypes = [vp, sz, vp]
        l.oracle_project_oxford.restype = None
        l.oracle_project_kitti.argtypes = [vp, sz, vp]
        l.oracle_project_kitti.restype = None
        _lib = l
    return _lib


def sensor_from_params(p) -> OracleSensor:
    return OracleSensor(p.horizon_scan, p.n_scan, p.ground_upper_scan, p.height_res)


def sensor_kind(kind: int) -> OracleSensor:
    s = OracleSensor()
    assert lib().oracle_sensor_params(kind, C.byref(s)) == 0
    return s


def belonging_grid(x, y):
    r, c = C.c_int(), C.c_int()
    lib().oracle_belonging_grid(float(np.float32(x)), float(np.float32(y)), C.byref(r), C.byref(c))
    return r.value, c.value


def order_cloud(sp: OracleSensor, pts: np.ndarray) -> np.ndarray:
    pts = np.ascontiguousarray(pts, dtype=POINT_DTYPE)
    out = np.empty(sp.n_scan * sp.horizon_scan, dtype=POINT_DTYPE)
    lib().oracle_order_cloud(C.byref(sp), pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def mark_ground(sp: OracleSensor, ordered: np.ndarray, angle_variant: int = 0):
    """angle_variant 0: float overloads of sqrt / atan2 / abs (adopted); 1: the double reading (bev_oracle.h)."""
    cloud = np.array(ordered, dtype=POINT_DTYPE, copy=True)
    gm = np.empty((sp.n_scan, sp.horizon_scan), dtype=np.int8)
    avg = np.empty(75 * 50, dtype=np.float32)
    lib().oracle_mark_ground_variant(C.byref(sp), cloud.ctypes.data, gm.ctypes.data, avg.ctypes.data, angle_variant)
    return cloud, gm, avg


def multi_bev(sp: OracleSensor, cloud: np.ndarray, interval: float = 1.0) -> np.ndarray:
    cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
    M = int(np.float32(224) / np.float32(interval))
    out = np.empty((24, M, M), dtype=np.uint8)
    lib().oracle_multi_bev(C.byref(sp), cloud.ctypes.data, len(cloud), interval, out.ctypes.data)
    return out


def single_bev(cloud: np.ndarray, interval: float = 1.0) -> np.ndarray:
    cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
    M = int(np.float32(224) / np.float32(interval))
    out = np.empty((M, M), dtype=np.uint8)
    lib().oracle_single_bev(cloud.ctypes.data, len(cloud), interval, out.ctypes.data)
    return out


def process_frame(sp: OracleSensor, pts: np.ndarray, want_gm: bool = True):
    pts = np.ascontiguousarray(pts, dtype=POINT_DTYPE)
    S = sp.n_scan * sp.horizon_scan
    ordered = np.empty(S, dtype=POINT_DTYPE)
    gm = np.empty((sp.n_scan, sp.horizon_scan), dtype=np.int8) if want_gm else None
    multi = np.empty((24, 224, 224), dtype=np.uint8)
    single = np.empty((224, 224), dtype=np.uint8)
    lib().oracle_process_frame(C.byref(sp), pts.ctypes.data, len(pts), ordered.ctypes.data,
                               gm.ctypes.data if want_gm else None, multi.ctypes.data, single.ctypes.data)
    return ordered, gm, multi, single


def float_bev(cloud: np.ndarray, interval: float = 1.0, skip_label0: bool = True) -> np.ndarray:
    cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
    M = int(np.float32(np.float32(200) / np.float32(interval)) + np.float32(1))
    out = np.empty((M, M), dtype=np.float32)
    lib().oracle_float_bev(cloud.ctypes.data, len(cloud), interval, 1 if skip_label0 else 0, out.ctypes.data)
    return out


def yaw_translate_matrix(tx: float, ty: float, tz: float, yaw_deg: float) -> np.ndarray:
    m = np.empty(12, dtype=np.float32)
    lib().oracle_yaw_translate_matrix(tx, ty, tz, yaw_deg, m.ctypes.data)
    return m


def transform_cloud(cloud: np.ndarray, m: np.ndarray) -> np.ndarray:
    cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
    m = np.ascontiguousarray(m, dtype=np.float32)
    out = np.empty_like(cloud)
    lib().oracle_transform_cloud(cloud.ctypes.data, len(cloud), m.ctypes.data, out.ctypes.data)
    return out


def project(kind: int, xyzi: np.ndarray) -> np.ndarray:
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32)
    n = xyzi.size // 4
    out = np.empty(64 * 2083 if kind == 2 else n, dtype=POINT_DTYPE)
    (lib().oracle_project_mulran, lib().oracle_project_oxford, lib().oracle_project_kitti)[kind](
        xyzi.ctypes.data, n, out.ctypes.data)
    return out
