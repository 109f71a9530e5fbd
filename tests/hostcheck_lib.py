"""ctypes loader for tests/hostcheck/libhostcheck.so (host build of csrc/bev_exact.h)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from bev_amd import POINT_DTYPE, BevParams

SO = Path(__file__).resolve().parent / "hostcheck" / "libhostcheck.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(str(SO))
        vp = C.c_void_p
        l.hc_angle.argtypes = [vp, vp, vp, C.c_size_t, vp]
        l.hc_angle.restype = None
        l.hc_tan_threshold_bits.restype = C.c_uint32
        l.hc_ground_cell.argtypes = [C.c_float, C.c_float]
        l.hc_bev_code.argtypes = [C.POINTER(BevParams), C.c_float, C.c_float, C.c_float, C.c_int]
        l.hc_bev_code.restype = C.c_uint32
        l.hc_count_advance_check.argtypes = [C.c_uint32]
        l.hc_count_advance_check.restype = C.c_uint64
        l.hc_small_div_check.argtypes = []
        l.hc_small_div_check.restype = C.c_uint64
        l.hc_angle_nodiv_check.argtypes = [C.c_uint64]
        l.hc_angle_nodiv_check.restype = C.c_uint64
        l.hc_exact_reciprocal_check.argtypes = [C.c_uint64]
        l.hc_exact_reciprocal_check.restype = C.c_uint64
        l.hc_key_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        l.hc_key_stats.restype = None
        l.hc_key_roundtrip.argtypes = [C.POINTER(BevParams), C.c_float, C.c_float, C.c_float, C.c_int]
        l.hc_process_frame.argtypes = [C.POINTER(BevParams), vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
        l.hc_process_frame.restype = None
        l.hc_libm_vs_host.argtypes = [C.c_uint64, vp]
        l.hc_libm_vs_host.restype = None
        l.hc_project.argtypes = [C.c_int, vp, C.c_uint32, vp]
        l.hc_project.restype = None
        l.hc_project_kitti.argtypes = [vp, C.c_uint32, vp]
        l.hc_project_kitti.restype = None
        l.hc_kitti_ring_min.argtypes = [vp]
        l.hc_kitti_ring_min.restype = None
        l.hc_select_major_frames.argtypes = [vp, C.c_uint32, vp]
        l.hc_select_major_frames.restype = C.c_uint32
        l.hc_keyframe_labels.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp]
        l.hc_keyframe_labels.restype = None
        l.hc_exhaustive_exact_forms.argtypes = [vp]
        l.hc_exhaustive_exact_forms.restype = None
        l.hc_derive_angle_threshold.argtypes = [vp]
        l.hc_derive_angle_threshold.restype = None
        l.hc_angle_vs_libm.argtypes = [C.c_uint64, C.c_uint64]
        l.hc_angle_vs_libm.restype = C.c_uint64
        l.hc_pcd_load.argtypes = [C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        l.hc_pcd_save.argtypes = [C.c_char_p, vp, C.c_size_t]
        l.hc_png_write.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
        l.hc_csv_u8.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
        l.hc_csv_u8.restype = C.c_size_t
        _lib = l
    return _lib


def pcd_load(path, cap=1 << 22):
    """host/FileFormats.cpp loadPCDFile -> (rc, points, width, height)"""
    out = np.zeros(cap, POINT_DTYPE)
    n, w, h = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().hc_pcd_load(str(path).encode(), out.ctypes.data, cap, C.byref(n), C.byref(w), C.byref(h))
    return rc, out[:min(n.value, cap)].copy(), w.value, h.value


def pcd_save(path, pts):
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    return lib().hc_pcd_save(str(path).encode(), pts.ctypes.data, len(pts))


def png_write(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().hc_png_write(str(path).encode(), img.ctypes.data, img.shape[0], img.shape[1])


def csv_u8(img):
    img = np.ascontiguousarray(img, np.uint8)
    n = lib().hc_csv_u8(img.ctypes.data, img.shape[0], img.shape[1], None, 0)
    buf = C.create_string_buffer(n)
    lib().hc_csv_u8(img.ctypes.data, img.shape[0], img.shape[1], buf, n)
    return buf.raw.decode()


def angle(dx, dy, dz):
    dx = np.ascontiguousarray(dx, np.float32)
    dy = np.ascontiguousarray(dy, np.float32)
    dz = np.ascontiguousarray(dz, np.float32)
    out = np.empty(len(dx), np.uint8)
    lib().hc_angle(dx.ctypes.data, dy.ctypes.data, dz.ctypes.data, len(dx), out.ctypes.data)
    return out


def process_frame(p: BevParams, pts):
    pts = np.ascontiguousarray(pts, dtype=POINT_DTYPE)
    S, M, L = p.slots, p.mat_size, p.n_layers
    ordered = np.empty(S, POINT_DTYPE)
    gm_a = np.empty((p.n_scan, p.horizon_scan), np.int8)
    gm = np.empty((p.n_scan, p.horizon_scan), np.int8)
    avg = np.empty(75 * 50, np.float32)
    multi = np.empty((L, M, M), np.uint8)
    single = np.empty((M, M), np.uint8)
    lib().hc_process_frame(C.byref(p), pts.ctypes.data, len(pts), ordered.ctypes.data, gm_a.ctypes.data,
                           gm.ctypes.data, avg.ctypes.data, multi.ctypes.data, single.ctypes.data)
    return ordered, gm, avg, multi, single


def project(kind: int, xyzi):
    xyzi = np.ascontiguousarray(xyzi, np.float32)
    n = xyzi.size // 4
    if kind == 2:
        out = np.empty(64 * 2083, POINT_DTYPE)
        lib().hc_project_kitti(xyzi.ctypes.data, n, out.ctypes.data)
        return out
    out = np.empty(n, POINT_DTYPE)
    lib().hc_project(kind, xyzi.ctypes.data, n, out.ctypes.data)
    return out


def select_major_frames(xyz):
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.empty(len(xyz), np.int32)
    m = lib().hc_select_major_frames(xyz.ctypes.data, len(xyz), out.ctypes.data)
    return out[:m].copy()


def keyframe_labels(xyz, major):
    xyz = np.ascontiguousarray(xyz, np.float32)
    major = np.ascontiguousarray(major, np.int32)
    out = np.empty((len(xyz), len(major)), np.float32)
    lib().hc_keyframe_labels(xyz.ctypes.data, len(xyz), major.ctypes.data, len(major), out.ctypes.data)
    return out


def key_stats(reset=True):
    """(candidate codes rebuilt from key + height, candidates that took the escape path) in hc_process_frame so far"""
    d, e = C.c_uint64(), C.c_uint64()
    lib().hc_key_stats(C.byref(d), C.byref(e), 1 if reset else 0)
    return int(d.value), int(e.value)


def key_roundtrip(p: BevParams, x, y, z, label) -> int:
    """0: wrong, 1: key + height reproduce bev_code (or flag 'no code'), 2: escape (the point itself is read)"""
    return int(lib().hc_key_roundtrip(C.byref(p), float(x), float(y), float(z), int(label)))
