#!/usr/bin/env python3
"""Every workgroup of the pipeline's kernels during ONE pass over the batch, on a time line (developer build: make -C
point-cloud-preprocessing-tools_amd tl).  Who shares the chip with whom, how many workgroups are resident, where the slots stand empty.
   BEV_AMD_LIB=.../csrc/libbev_tl_all.so python3 scripts/pipeline_timeline.py [frames] [HDL_64E|OS1_64] [bin us] [sub-batch]"""
import os, sys, ctypes as C, collections
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd'))
import torch
import bev_amd
from bev_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sensor = sys.argv[2] if len(sys.argv) > 2 else "HDL_64E"
BIN = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
SB = int(sys.argv[4]) if len(sys.argv) > 4 else 500   # sub-batch (the context's max_batch), as in bench.py
p = bev_amd.params_for_sensor(sensor)
S, M, L = p.slots, p.mat_size, p.n_layers
gen = (lambda i: synth.sweep(p, i, keep=0.98, n_dup=5000)) if sensor == "HDL_64E" else (lambda i: synth.firing_order(p, i))
frames = [gen(i) for i in range(min(F, 32))]
frames = [frames[i % len(frames)] for i in range(F)]
counts = np.array([len(f) for f in frames]); offsets = np.zeros(F + 1, dtype=np.uint64); offsets[1:] = np.cumsum(counts)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
d_o = torch.empty(F * S * 32, dtype=torch.uint8, device=dev); d_m = torch.empty(F * L * M * M, dtype=torch.uint8, device=dev)
d_s = torch.empty(F * M * M, dtype=torch.uint8, device=dev)
ctx = bev_amd.BevContext(p, device=0, max_batch=min(F, SB), max_points=int(counts.max()))
lib = bev_amd.load_lib()
lib.bev_tl_all.argtypes = [C.c_void_p, C.c_int, C.c_int]
for _ in range(4):
    ctx.process_device(F, d_in.data_ptr(), offsets, d_o.data_ptr(), d_m.data_ptr(), d_s.data_ptr())
    ctx.synchronize()
lib.bev_tl_all(None, 0, 1)
REP = int(os.environ.get("TL_CALLS", "3"))  # back-to-back calls: the later stages of a call's last sub-batches ride in the next call's launches
for _ in range(REP):
    ctx.process_device(F, d_in.data_ptr(), offsets, d_o.data_ptr(), d_m.data_ptr(), d_s.data_ptr())
ctx.synchronize()
cap = 1 << 19
buf = (C.c_longlong * (cap * 4))()
n = lib.bev_tl_all(buf, cap, 1)
rec = np.frombuffer(buf, dtype=np.int64).reshape(cap, 4)[:n]
t0 = rec[:, 0].min()
st = (rec[:, 0] - t0) / 100.0; en = (rec[:, 1] - t0) / 100.0; life = en - st
kid = (rec[:, 2] >> 40) & 0xff
hw = rec[:, 2] & 0xffffffff; xcc = (rec[:, 2] >> 32) & 0xf
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
names = {1: "walk", 2: "cell_sums", 3: "resolve", 4: "raster", 12: "probe"}
span = en.max()
print(f"{sensor}: {REP} calls of {F} frames, {n} workgroups in {span:.1f} us ({REP * F / span * 1e6:.0f} frames/s); {len(set(cu.tolist()))} CUs seen")
for k, nm in names.items():
    m = kid == k
    if m.any():
        print(f"  {nm:9s} n {m.sum():6d} lifetime median {np.median(life[m]):7.1f} p10 {np.percentile(life[m], 10):7.1f} p90 {np.percentile(life[m], 90):7.1f} us; "
              f"workgroup-time {life[m].sum() / 1e3:8.1f} ms = {life[m].sum() / span:7.1f} resident on average; first start {st[m].min():.0f} last end {en[m].max():.0f}")
T = int(span // BIN) + 1
mid = (np.arange(T) + 0.5) * BIN
print(f"resident workgroups every {BIN:.0f} us (walk / cell_sums / resolve / raster / probe | all):")
rows = []
for k in names:
    m = kid == k
    s_, e_ = np.sort(st[m]), np.sort(en[m])
    rows.append(np.searchsorted(s_, mid, side="right") - np.searchsorted(e_, mid, side="right"))
rows = np.array(rows)
for i in range(T):
    print(f"  {mid[i]:7.0f} us: " + " ".join(f"{int(v):5d}" for v in rows[:, i]) + f" | {int(rows[:, i].sum()):5d}")
# how long a workgroup of the back stage waits for its launch's first one: dispatch order = block order
for k in (2, 4):
    m = kid == k
    # launches are told apart by gaps in the start times of block 0
    b0 = np.sort(st[m & (rec[:, 3] == 0)])
    print(f"  {names[k]}: launches start at", [round(float(x)) for x in b0])
