"""Shared by the CPU and GPU golden tests: rebuild the fixture inputs and hash outputs."""
import hashlib
from pathlib import Path

import numpy as np

import bev_amd
from bev_amd import synth

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_input(p, sensor, name):
    if name == "sweep0":
        return synth.sweep(p, 0)
    if name == "sweep1":
        return synth.sweep(p, 1)
    if name == "firing1":
        return synth.firing_order(p, 1)
    if name == "adv3":
        return synth.adversarial(p, 20000, 3, True)
    if name == "config1_16k":
        return synth.sweep_unique(p, 0, 16384)
    if name == "concat6":
        return synth.concat(p, 0, n_sweeps=6)
    raise KeyError(name)


def summarize(ordered, gm, avg, multi, single):
    d = {"ordered": sha(ordered), "labels": sha(ordered["label"]), "multi_bin": sha(multi), "single": sha(single),
         "occupied_multi": int((multi == 255).sum()), "single_sum": int(single.astype(np.int64).sum())}
    if gm is not None:
        d["ground_mat"] = sha(gm)
        d["ground_slots"] = int((gm == 1).sum())
    if avg is not None:
        d["cell_avg"] = sha(avg)
    return d


def load_tiny():
    z = np.load(GOLDEN_DIR / "tiny_hdl32.npz")
    p = bev_amd.params_for_sensor("HDL_32E")
    ordered = np.zeros(p.slots, bev_amd.POINT_DTYPE)
    ordered[z["ordered_idx"]] = z["ordered_pts"].view(bev_amd.POINT_DTYPE) if z["ordered_pts"].dtype != bev_amd.POINT_DTYPE else z["ordered_pts"]
    multi = np.zeros((24, 224, 224), np.uint8)
    idx = z["multi_idx"].astype(np.int64)
    multi[idx[:, 0], idx[:, 1], idx[:, 2]] = 255
    pts = z["points"]
    if pts.dtype != bev_amd.POINT_DTYPE:
        pts = pts.view(bev_amd.POINT_DTYPE)
    return p, np.ascontiguousarray(pts), ordered, z["ground_mat"], multi, z["single"]
