"""CPU: the C-ABI library loads, exports every symbol include/bev_mi355x.h declares,
answers the host-only queries, and FAILS LOUDLY when no GPU is present."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

import bev_amd

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "bev_mi355x.h"


def _declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(bev_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = bev_amd.load_lib()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in the header but not exported"
    assert sorted(bev_amd.ABI_SYMBOLS) == names
    assert lib.bev_abi_version() == 1


def test_exports_have_c_linkage():
    out = subprocess.run(["nm", "-D", "--defined-only", str(bev_amd.LIB_PATH)], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in _declared_functions():
        assert n in exported


def test_sensor_table_and_sizes():
    # src/Utility.cpp:96-118 + BatchMultiBevGen.cpp:266-269
    expect = {"HDL_32E": (32, 1056, 20, 0.5), "HDL_64E": (64, 2083, 50, 0.25), "OS1_64": (64, 1024, 31, 1.0)}
    lib = bev_amd.load_lib()
    for name, (n, h, g, res) in expect.items():
        p = bev_amd.params_for_sensor(name)
        assert (p.n_scan, p.horizon_scan, p.ground_upper_scan, p.height_res) == (n, h, g, res)
        assert (p.interval, p.max_range, p.n_layers, p.lidar_to_ground) == (1.0, 112, 24, 2.0)
        assert lib.bev_num_slots(C.byref(p)) == n * h
        assert lib.bev_multi_bytes(C.byref(p)) == 24 * 224 * 224 == 1204224   # the .bin size
        assert lib.bev_single_bytes(C.byref(p)) == 224 * 224
    # substring match like parseSensorType (src/Utility.cpp:74-83)
    assert bev_amd.params_for_sensor("kitti_HDL_64E_raw").horizon_scan == 2083
    with pytest.raises(bev_amd.BevError):
        bev_amd.params_for_sensor("VLP_16")  # the reference leaves the struct uninitialised; we reject


def test_invalid_params_rejected_before_touching_the_gpu():
    lib = bev_amd.load_lib()
    p = bev_amd.params_for_sensor("HDL_32E")
    p.interval = 0.3  # MAT_SIZE 746: not a multiple of 16, above 512
    assert lib.bev_multi_bytes(C.byref(p)) == 0
    h = C.c_void_p()
    assert lib.bev_create(C.byref(h), 0, C.byref(p), 1, 1000) in (-5, -1)
    p = bev_amd.params_for_sensor("HDL_32E")
    p.ground_upper_scan = 31  # lo = 1: the reference itself would index out of bounds
    assert lib.bev_create(C.byref(h), 0, C.byref(p), 1, 1000) == -1
    assert lib.bev_create(C.byref(h), 0, None, 1, 1000) == -1


def test_no_gpu_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = bev_amd.params_for_sensor("HDL_32E")
    with pytest.raises(bev_amd.BevError, match="no usable HIP device"):
        bev_amd.BevContext(p, device=0, max_batch=1, max_points=1000)
    assert b"no CPU path" in bev_amd.load_lib().bev_strerror(-2)
