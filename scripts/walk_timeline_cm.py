#!/usr/bin/env python3
"""Lifetimes of the workgroups of ONE firing-order walk launch, by strip (developer build: make clk).
   BEV_AMD_LIB=.../libbev_mi355x_clk.so python3 scripts/walk_timeline_cm.py [frames] [real]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd'))
os.environ.setdefault("BEV_LANES", "1")
import torch
import bev_amd
from bev_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
real = len(sys.argv) > 2
p = bev_amd.params_for_sensor("OS1_64")
S, M, L = p.slots, p.mat_size, p.n_layers
frames = [(synth.firing_real(p, i) if real else synth.firing_order(p, i)) for i in range(min(F, 32))]
frames = [frames[i % len(frames)] for i in range(F)]
counts = np.array([len(f) for f in frames]); offsets = np.zeros(F + 1, dtype=np.uint64); offsets[1:] = np.cumsum(counts)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
d_o = torch.empty(F * S * 32, dtype=torch.uint8, device=dev); d_m = torch.empty(F * L * M * M, dtype=torch.uint8, device=dev)
d_s = torch.empty(F * M * M, dtype=torch.uint8, device=dev)
ctx = bev_amd.BevContext(p, device=0, max_batch=F, max_points=int(counts.max()))
for _ in range(3):
    ctx.process_device(F, d_in.data_ptr(), offsets, d_o.data_ptr(), d_m.data_ptr(), d_s.data_ptr())
    ctx.synchronize()
lib = bev_amd.load_lib()
cap = 8192
buf = (C.c_longlong * (cap * 4))()
n = lib.bev_clk_walk_timeline(buf, cap)
strips = (p.horizon_scan + 235) // 236
rec = np.frombuffer(buf, dtype=np.int64).reshape(cap, 4)[: min(n, 8 * ((F + 7) // 8) * strips)]
blk = np.arange(len(rec)); ok = rec[:, 1] > 0
strip = strips - 1 - ((blk >> 3) % strips)
t0 = rec[ok, 0].min()
life = (rec[:, 1] - rec[:, 0]) / 100.0; start = (rec[:, 0] - t0) / 100.0
print(f"{ok.sum()} workgroups recorded; launch spans {((rec[ok,1]-t0)/100.0).max():.1f} us")
for s in range(strips):
    m = ok & (strip == s)
    print(f"strip {s}: n {m.sum()} lifetime us median {np.median(life[m]):.1f} mean {life[m].mean():.1f} p90 {np.percentile(life[m], 90):.1f}; start median {np.median(start[m]):.1f}")
