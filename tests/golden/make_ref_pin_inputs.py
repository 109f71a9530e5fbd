#!/usr/bin/env python3
"""Writes the inputs tests/golden/ref_dump.cpp reads: tests/golden/ref_pin/<name>.points = raw 32-byte records of this
repo's deterministic synthetic frames (small ones: the files are committed).  Run from the repo root:
    python tests/golden/make_ref_pin_inputs.py"""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd"))
sys.path.insert(0, str(REPO / "tests"))
import bev_amd  # noqa: E402
import golden_util as gu  # noqa: E402
from bev_amd import synth  # noqa: E402

out = HERE / "ref_pin"
out.mkdir(exist_ok=True)
p32 = bev_amd.params_for_sensor("HDL_32E")
_, tiny, *_ = gu.load_tiny()                       # 2,085 points: duplicates, out-of-range rows / cols, intensity -1
cases = {"tiny_hdl32": tiny, "config1_16k": synth.sweep_unique(p32, 0, 16384)}   # BASELINE.json configs[0]
for name, pts in cases.items():
    (out / f"{name}.points").write_bytes(pts.tobytes())
    print(name, len(pts), "points")
