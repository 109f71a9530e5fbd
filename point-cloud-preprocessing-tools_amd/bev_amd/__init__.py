"""ctypes binding of the C ABI in include/bev_mi355x.h (libbev_mi355x.so).

Python is plumbing here (tests, bench.py, smoke): the product is the HIP
library.  There is no Python or CPU implementation of the hot path in this
package; if the library is missing or no GPU is usable, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent
REPO_DIR = PKG_DIR.parent
# BEV_AMD_LIB: developer override (e.g. the `make clk` build with in-kernel phase clocks); never a fallback
LIB_PATH = Path(os.environ["BEV_AMD_LIB"]) if os.environ.get("BEV_AMD_LIB") else PKG_DIR / "csrc" / "libbev_mi355x.so"
SYNTH_PATH = PKG_DIR / "synth" / "libbev_synth.so"

# pcl::PointXYZIRCT in memory (reference BatchMultiBevGen.h:43-54), 32 bytes
POINT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "z", "_pad0", "intensity", "row", "col", "t", "label", "_pad1"],
        "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<u2", "<u2", "<u4", "<i2", "<u2"],
        "offsets": [0, 4, 8, 12, 16, 20, 22, 24, 28, 30],
        "itemsize": 32,
    }
)

GROUND_GRID_CELLS = 75 * 50


class BevParams(C.Structure):
    _fields_ = [
        ("n_scan", C.c_int32),
        ("horizon_scan", C.c_int32),
        ("ground_upper_scan", C.c_int32),
        ("height_res", C.c_float),
        ("interval", C.c_float),
        ("max_range", C.c_int32),
        ("n_layers", C.c_int32),
        ("lidar_to_ground", C.c_float),
    ]

    @property
    def slots(self) -> int:
        return self.n_scan * self.horizon_scan

    @property
    def mat_size(self) -> int:
        return int(np.float32(self.max_range * 2) / np.float32(self.interval))


class KernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("launches", C.c_uint64),
        ("total_ms", C.c_double),
        ("frames", C.c_uint64),
    ]


class BevError(RuntimeError):
    pass


_lib = None
LAYOUT_UNKNOWN, LAYOUT_STRUCTURED, LAYOUT_FIRING_ORDER = 0, 3, 4  # bev_set_layout_hint

# every symbol include/bev_mi355x.h declares
ABI_SYMBOLS = [
    "bev_params_for_sensor", "bev_num_slots", "bev_multi_bytes", "bev_single_bytes",
    "bev_create", "bev_destroy", "bev_strerror", "bev_last_error",
    "bev_process_batch", "bev_process_device_resident", "bev_synchronize",
    "bev_order_cloud", "bev_mark_ground", "bev_multi_bev", "bev_single_bev",
    "bev_float_bev", "bev_float_bev_size", "bev_transform_cloud", "bev_yaw_translate_matrix", "bev_project_xyzi", "bev_project_out_points", "bev_host_alloc", "bev_host_free",
    "bev_set_lanes", "bev_set_layout_hint", "bev_profile_enable", "bev_profile_reset", "bev_profile_get",
    "bev_debug_get_cell_avg", "bev_debug_get_frame_info", "bev_debug_get_code_overflow", "bev_debug_angle_predicate", "bev_abi_version",
]


def load_lib() -> C.CDLL:
    """Load libbev_mi355x.so; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two
    # HIP runtimes in one process cannot both open the GPU.  Importing torch FIRST
    # makes the dynamic linker resolve our NEEDED libamdhip64.so.7 to the copy torch
    # already loaded (same SONAME), so the process has exactly one runtime.
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not LIB_PATH.exists():
        raise BevError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() or "
            f"`make -C {PKG_DIR}`. There is no CPU/Python fallback for the hot path."
        )
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    lib.bev_params_for_sensor.argtypes = [C.c_char_p, C.POINTER(BevParams)]
    lib.bev_num_slots.argtypes = [C.POINTER(BevParams)]
    lib.bev_num_slots.restype = sz
    lib.bev_multi_bytes.argtypes = [C.POINTER(BevParams)]
    lib.bev_multi_bytes.restype = sz
    lib.bev_single_bytes.argtypes = [C.POINTER(BevParams)]
    lib.bev_single_bytes.restype = sz
    lib.bev_create.argtypes = [C.POINTER(vp), i32, C.POINTER(BevParams), i32, sz]
    lib.bev_destroy.argtypes = [vp]
    lib.bev_destroy.restype = None
    lib.bev_strerror.argtypes = [i32]
    lib.bev_strerror.restype = C.c_char_p
    lib.bev_last_error.argtypes = [vp]
    lib.bev_last_error.restype = C.c_char_p
    lib.bev_process_batch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(vp),
                                      C.POINTER(vp), C.POINTER(vp)]
    lib.bev_process_device_resident.argtypes = [vp, i32, vp, C.POINTER(C.c_uint64), vp, vp, vp, vp]
    lib.bev_synchronize.argtypes = [vp]
    lib.bev_order_cloud.argtypes = [vp, vp, u32, vp]
    lib.bev_mark_ground.argtypes = [vp, vp, vp]
    lib.bev_multi_bev.argtypes = [vp, vp, u32, vp]
    lib.bev_single_bev.argtypes = [vp, vp, u32, vp]
    lib.bev_float_bev.argtypes = [vp, vp, u32, C.c_float, i32, vp]
    lib.bev_float_bev_size.argtypes = [C.c_float]
    lib.bev_float_bev_size.restype = sz
    lib.bev_project_xyzi.argtypes = [vp, i32, vp, u32, vp]
    lib.bev_transform_cloud.argtypes = [vp, vp, u32, vp, vp]
    lib.bev_yaw_translate_matrix.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib.bev_yaw_translate_matrix.restype = None
    lib.bev_host_alloc.argtypes = [C.POINTER(vp), sz]
    lib.bev_host_free.argtypes = [vp]
    lib.bev_project_out_points.argtypes = [i32, u32]
    lib.bev_project_out_points.restype = C.c_size_t
    lib.bev_set_lanes.argtypes = [vp, i32]
    try:  # (an older build of the library, selected with BEV_AMD_LIB for a same-box A/B: scripts/ab_libs.sh)
        lib.bev_set_layout_hint.argtypes = [vp, i32]
    except AttributeError:
        pass
    lib.bev_profile_enable.argtypes = [vp, i32]
    lib.bev_profile_reset.argtypes = [vp]
    lib.bev_profile_get.argtypes = [vp, C.POINTER(KernelStat), i32]
    lib.bev_debug_get_cell_avg.argtypes = [vp, i32, i32, vp]
    lib.bev_debug_get_frame_info.argtypes = [vp, i32, i32, vp]
    if hasattr(lib, "bev_debug_get_code_overflow"):  # (absent from older builds selected through BEV_AMD_LIB for A/B runs)
        lib.bev_debug_get_code_overflow.argtypes = [vp, i32, i32, vp]
    lib.bev_debug_angle_predicate.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.bev_abi_version.restype = i32
    _lib = lib
    return lib


def params_for_sensor(sensor: str) -> BevParams:
    p = BevParams()
    rc = load_lib().bev_params_for_sensor(sensor.encode(), C.byref(p))
    if rc != 0:
        raise BevError(f"unknown sensor type {sensor!r}")
    return p


def _ptr(a: np.ndarray | None) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(None)


class BevContext:
    """One context per GPU (bev_create / bev_destroy)."""

    def __init__(self, params: BevParams, device: int = 0, max_batch: int = 8, max_points: int | None = None):
        self.lib = load_lib()
        self.params = params
        self.S = params.slots
        self.M = params.mat_size
        self.L = params.n_layers
        self.max_batch = max_batch
        self.max_points = int(max_points if max_points is not None else self.S + 8192)
        self._h = C.c_void_p(None)
        rc = self.lib.bev_create(C.byref(self._h), device, C.byref(params), max_batch, self.max_points)
        if rc != 0:
            raise BevError(f"bev_create failed: {self.lib.bev_strerror(rc).decode()} (status {rc})")

    def close(self):
        if self._h:
            self.lib.bev_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.bev_strerror(rc).decode()
            detail = self.lib.bev_last_error(self._h).decode() if self._h else ""
            raise BevError(f"{what} failed: {msg} (status {rc}) {detail}")

    # ---- whole hot path, host buffers ---------------------------------
    def process_batch(self, frames, want_multi=True, want_single=True, want_ground_mat=False):
        n = len(frames)
        frames = [np.ascontiguousarray(f, dtype=POINT_DTYPE) for f in frames]
        ordered = np.empty((n, self.S), dtype=POINT_DTYPE)
        multi = np.empty((n, self.L, self.M, self.M), dtype=np.uint8) if want_multi else None
        single = np.empty((n, self.M, self.M), dtype=np.uint8) if want_single else None
        gm = np.empty((n, self.params.n_scan, self.params.horizon_scan), dtype=np.int8) if want_ground_mat else None
        VP = C.c_void_p * max(n, 1)
        pts = VP(*[f.ctypes.data if len(f) else None for f in frames])
        npts = (C.c_uint32 * max(n, 1))(*[len(f) for f in frames])
        o = VP(*[ordered[i].ctypes.data for i in range(n)])
        m = VP(*[multi[i].ctypes.data for i in range(n)]) if want_multi else None
        s = VP(*[single[i].ctypes.data for i in range(n)]) if want_single else None
        g = VP(*[gm[i].ctypes.data for i in range(n)]) if want_ground_mat else None
        rc = self.lib.bev_process_batch(self._h, n, pts, npts, o, m, s, g)
        self._check(rc, "bev_process_batch")
        return ordered, multi, single, gm

    # ---- whole hot path, device pointers --------------------------------
    def process_device(self, n_frames, d_pts, offsets, d_ordered, d_multi, d_single, d_ground_mat=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert offsets.shape[0] == n_frames + 1
        rc = self.lib.bev_process_device_resident(
            self._h, n_frames, C.c_void_p(d_pts), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
            C.c_void_p(d_ordered), C.c_void_p(d_multi), C.c_void_p(d_single), C.c_void_p(d_ground_mat))
        self._check(rc, "bev_process_device_resident")

    def synchronize(self):
        self._check(self.lib.bev_synchronize(self._h), "bev_synchronize")

    # ---- per-function entry points ---------------------------------------
    def order_cloud(self, pts):
        pts = np.ascontiguousarray(pts, dtype=POINT_DTYPE)
        out = np.empty(self.S, dtype=POINT_DTYPE)
        self._check(self.lib.bev_order_cloud(self._h, _ptr(pts) if len(pts) else None, len(pts), _ptr(out)),
                    "bev_order_cloud")
        return out

    def mark_ground(self, ordered, want_ground_mat=True):
        cloud = np.array(ordered, dtype=POINT_DTYPE, copy=True)
        assert cloud.shape == (self.S,)
        gm = np.empty((self.params.n_scan, self.params.horizon_scan), dtype=np.int8) if want_ground_mat else None
        self._check(self.lib.bev_mark_ground(self._h, _ptr(cloud), _ptr(gm)), "bev_mark_ground")
        return cloud, gm

    def multi_bev(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
        out = np.empty((self.L, self.M, self.M), dtype=np.uint8)
        self._check(self.lib.bev_multi_bev(self._h, _ptr(cloud) if len(cloud) else None, len(cloud), _ptr(out)),
                    "bev_multi_bev")
        return out

    def single_bev(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
        out = np.empty((self.M, self.M), dtype=np.uint8)
        self._check(self.lib.bev_single_bev(self._h, _ptr(cloud) if len(cloud) else None, len(cloud), _ptr(out)),
                    "bev_single_bev")
        return out

    def float_bev(self, cloud, interval=1.0, skip_label0=True):
        cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
        M = int(self.lib.bev_float_bev_size(interval))
        out = np.empty((M, M), dtype=np.float32)
        self._check(self.lib.bev_float_bev(self._h, _ptr(cloud) if len(cloud) else None, len(cloud), interval,
                                           1 if skip_label0 else 0, _ptr(out)), "bev_float_bev")
        return out

    def transform_cloud(self, cloud, m):
        cloud = np.ascontiguousarray(cloud, dtype=POINT_DTYPE)
        m = np.ascontiguousarray(m, dtype=np.float32).reshape(12)
        out = np.empty_like(cloud)
        self._check(self.lib.bev_transform_cloud(self._h, _ptr(cloud) if len(cloud) else None, len(cloud), _ptr(m),
                                                 _ptr(out) if len(cloud) else None), "bev_transform_cloud")
        return out

    def project_xyzi(self, kind: int, xyzi):
        """kind 0: MulRan/Ouster (n, 4) interleaved; kind 1: Oxford (4, n) planes; kind 2: KITTI (n, 4)
        interleaved, returns the structured 64 * 2083 cloud."""
        xyzi = np.ascontiguousarray(xyzi, dtype=np.float32)
        n = xyzi.size // 4
        n_out = int(self.lib.bev_project_out_points(kind, n))
        out = np.empty(n_out, dtype=POINT_DTYPE)
        self._check(self.lib.bev_project_xyzi(self._h, kind, _ptr(xyzi) if n else None, n, _ptr(out) if n_out else None),
                    "bev_project_xyzi")
        return out

    def set_layout_hint(self, layout: int):
        """LAYOUT_UNKNOWN (the library looks), LAYOUT_STRUCTURED, LAYOUT_FIRING_ORDER: include/bev_mi355x.h"""
        self._check(self.lib.bev_set_layout_hint(self._h, layout), "bev_set_layout_hint")

    # ---- measurement / test hooks ------------------------------------------
    def set_lanes(self, n: int) -> int:
        r = self.lib.bev_set_lanes(self._h, n)
        if r < 0:
            self._check(r, "bev_set_lanes")
        return r

    def profile_enable(self, on=True):
        self._check(self.lib.bev_profile_enable(self._h, 1 if on else 0), "bev_profile_enable")

    def profile_reset(self):
        self._check(self.lib.bev_profile_reset(self._h), "bev_profile_reset")

    def profile_get(self):
        arr = (KernelStat * 16)()
        n = self.lib.bev_profile_get(self._h, arr, 16)
        if n < 0:
            self._check(n, "bev_profile_get")
        return [
            {"name": arr[i].name.decode(), "launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms),
             "frames": int(arr[i].frames)}
            for i in range(min(n, 16))
        ]

    def cell_avg(self, first_frame=0, n_frames=1):
        out = np.empty((n_frames, GROUND_GRID_CELLS), dtype=np.float32)
        self._check(self.lib.bev_debug_get_cell_avg(self._h, first_frame, n_frames, _ptr(out)),
                    "bev_debug_get_cell_avg")
        return out

    def frame_info(self, first_frame=0, n_frames=1):
        """(n, 4) uint32: T, mode (0 general, 1 sorted prefix read in place, 2 read in place then redone, 3 structured cloud
        read in place, 4 firing order read in place), consumed, failed of the frames of the last sub-batch."""
        out = np.empty((n_frames, 4), dtype=np.uint32)
        self._check(self.lib.bev_debug_get_frame_info(self._h, first_frame, n_frames, _ptr(out)), "bev_debug_get_frame_info")
        return out

    def code_overflow(self, first_frame=0, n_frames=1):
        """(n,) uint32: per frame of the last sub-batch, the number of (writer, raster band) code lists that overflowed;
        those bands were rastered from the ordered cloud instead"""
        if not hasattr(self.lib, "bev_debug_get_code_overflow"):
            raise BevError("this build of libbev_mi355x.so (selected through BEV_AMD_LIB?) has no bev_debug_get_code_overflow")
        out = np.empty(n_frames, dtype=np.uint32)
        self._check(self.lib.bev_debug_get_code_overflow(self._h, first_frame, n_frames, _ptr(out)), "bev_debug_get_code_overflow")
        return out

    def angle_predicate(self, dx, dy, dz):
        dx = np.ascontiguousarray(dx, dtype=np.float32)
        dy = np.ascontiguousarray(dy, dtype=np.float32)
        dz = np.ascontiguousarray(dz, dtype=np.float32)
        out = np.empty(dx.shape[0], dtype=np.uint8)
        self._check(self.lib.bev_debug_angle_predicate(self._h, _ptr(dx), _ptr(dy), _ptr(dz), _ptr(out), dx.shape[0]),
                    "bev_debug_angle_predicate")
        return out


def yaw_translate_matrix(tx: float, ty: float, tz: float, yaw_deg: float) -> np.ndarray:
    """[R | t] of cloud_manip (CloudManip.cpp:119-128) as 12 floats, row-major; host arithmetic only."""
    m = np.empty(12, dtype=np.float32)
    load_lib().bev_yaw_translate_matrix(tx, ty, tz, yaw_deg, _ptr(m))
    return m


def host_alloc(shape, dtype) -> np.ndarray:
    """A numpy array in page-locked host memory (bev_host_alloc); the memory is never returned to the system —
    meant for long-lived I/O buffers of a driver script."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    rc = load_lib().bev_host_alloc(C.byref(p), n)
    if rc != 0 or not p.value:
        raise BevError(f"bev_host_alloc({n}) failed: status {rc}")
    buf = (C.c_char * n).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def algorithmic_bytes_per_frame(params: BevParams, n_points: float) -> float:
    """32*P + 32*S + n_layers*M*M + M*M."""
    M = params.mat_size
    return 32.0 * n_points + 32.0 * params.slots + params.n_layers * M * M + M * M
