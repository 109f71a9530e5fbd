#!/usr/bin/env python3
"""When and where the workgroups of ONE in-place walk launch ran (developer build: make -C point-cloud-preprocessing-tools_amd clk).
   BEV_AMD_LIB=.../libbev_mi355x_clk.so python3 scripts/walk_timeline.py [frames]
Every workgroup records its start, end (100 MHz clock), HW_ID and XCC_ID in device memory; nothing is printed from the GPU."""
import os, sys, ctypes as C, collections
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd'))
os.environ.setdefault("BEV_LANES", "1")
import torch
import bev_amd
from bev_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
p = bev_amd.params_for_sensor("HDL_64E")
S, M, L = p.slots, p.mat_size, p.n_layers
frames = [synth.sweep(p, i, keep=0.98, n_dup=5000) for i in range(min(F, 16))]
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
rec = np.frombuffer(buf, dtype=np.int64).reshape(cap, 4)[: min(n, 8 * F)]
rec = rec[rec[:, 1] > 0]
t0 = rec[:, 0].min()
st = (rec[:, 0] - t0) / 100.0; en = (rec[:, 1] - t0) / 100.0
print(f"{len(rec)} workgroups; launch spans {en.max():.1f} us; lifetime us min {np.min(en - st):.1f} median {np.median(en - st):.1f} max {np.max(en - st):.1f}")
print("start offsets us (deciles):", [round(float(x), 1) for x in np.percentile(st, range(0, 101, 10))])
print("end   offsets us (deciles):", [round(float(x), 1) for x in np.percentile(en, range(0, 101, 10))])
hw = rec[:, 2]; xcc = rec[:, 3] & 0xf
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
per = collections.Counter(cu.tolist())
print("distinct CUs", len(per), "workgroups per CU: min", min(per.values()), "max", max(per.values()), "histogram", sorted(collections.Counter(per.values()).items()))
# concurrency: resident workgroups over time (10 us bins)
T = int(en.max() // 10) + 1
conc = [int(((st <= 10 * k + 5) & (en > 10 * k + 5)).sum()) for k in range(T)]
print("resident workgroups every 10 us:", conc)
life = en - st
slots = 4 * 256
print(f"sum of lifetimes {life.sum():.0f} us over {slots} slots = {life.sum() / slots:.1f} us of a {en.max():.1f} us launch: {life.sum() / slots / en.max():.3f} of the slot-time used")
strips = (p.horizon_scan + 235) // 236
blk = np.arange(len(hw)) if len(hw) == 8 * ((F + 7) // 8) * strips else None
if blk is not None:
    s_of = (blk >> 3) % strips
    for s in range(strips):
        m = s_of == s
        print(f"strip {s}: lifetime median {np.median(life[m]):.1f} p90 {np.percentile(life[m], 90):.1f}")
