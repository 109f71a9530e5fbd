#!/usr/bin/env python3
"""When, where and for how long every workgroup of ONE k_cell_sums launch ran (500 frames, one lane), by quarter, XCD and
number of candidates; the phase clocks of frame 12's four workgroups (developer build: make -C point-cloud-preprocessing-tools_amd cstl).
   BEV_AMD_LIB=.../csrc/libbev_cstl.so python3 scripts/cell_sums_timeline.py [OS1_64|HDL_64E]
This is the tool that found the second round of an OS1-64 launch waiting 55 us for the long quarter (round 5)."""
import os, sys, collections, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd'))
os.environ.setdefault("BEV_LANES", "1")
import torch, bev_amd
from bev_amd import synth
F=500; sensor=sys.argv[1] if len(sys.argv)>1 else "OS1_64"
p = bev_amd.params_for_sensor(sensor)
S, M, L = p.slots, p.mat_size, p.n_layers
mk = (lambda i: synth.firing_order(p,i)) if sensor=="OS1_64" else (lambda i: synth.sweep(p,i,keep=0.98,n_dup=5000))
frames=[mk(i) for i in range(32)]; frames=[frames[i%32] for i in range(F)]
counts=np.array([len(f) for f in frames]); offsets=np.zeros(F+1,dtype=np.uint64); offsets[1:]=np.cumsum(counts)
dev=torch.device("cuda",0)
d_in=torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
d_o=torch.empty(F*S*32,dtype=torch.uint8,device=dev); d_m=torch.empty(F*L*M*M,dtype=torch.uint8,device=dev); d_s=torch.empty(F*M*M,dtype=torch.uint8,device=dev)
ctx=bev_amd.BevContext(p,device=0,max_batch=F,max_points=int(counts.max()))
for _ in range(3):
    ctx.process_device(F,d_in.data_ptr(),offsets,d_o.data_ptr(),d_m.data_ptr(),d_s.data_ptr()); ctx.synchronize()
lib=bev_amd.load_lib(); cap=8192
buf=(C.c_longlong*(cap*4))(); n=lib.bev_clk_walk_timeline(buf,cap)
rec=np.frombuffer(buf,dtype=np.int64).reshape(cap,4)[:2016]; ok=rec[:,1]>0
rec=rec[ok]; ok=np.ones(len(rec),bool); cl=rec[:,0]>np.percentile(rec[:,0],30)-2000; rec=rec[cl]; ok=ok[cl]; blk0=np.arange(2016)[:len(cl)][cl]
t0=rec[ok,0].min(); st=(rec[ok,0]-t0)/100.0; en=(rec[ok,1]-t0)/100.0
print(sensor, ok.sum(),"wgs; span",round(en.max(),1),"us; lifetime median",round(float(np.median(en-st)),1),"p10",round(float(np.percentile(en-st,10)),1),"p90",round(float(np.percentile(en-st,90)),1))
print("start deciles",[round(float(x),1) for x in np.percentile(st,range(0,101,10))])
T=int(en.max()//10)+1
print("resident every 10us",[int(((st<=10*k+5)&(en>10*k+5)).sum()) for k in range(T)])
blk=blk0; life=en-st; xcc=(rec[ok,3]&0xf)
for x in range(8):
    m=xcc==x; m2=(blk%8)==x
    print(f"xcc {x}: n {m.sum()} lifetime median {np.median(life[m]):.1f} p90 {np.percentile(life[m],90):.1f} | block%8=={x}: n {m2.sum()} median {np.median(life[m2]):.1f} start median {np.median(st[m2]):.1f} first-round median life {np.median(life[m2&(st<st.min()+20)]):.1f} second-round start median {np.median(st[m2&(st>st.min()+20)]) if (m2&(st>st.min()+20)).any() else -1:.1f}")
fr = st < st.min() + 20
q = (rec[:, 3] >> 8) & 3
gc = rec[:, 3] >> 32
for k in range(4):
    m = fr & (q == k)
    print(f"first round, quarter {k}: n {m.sum()} life median {np.median(life[m]):.1f} p90 {np.percentile(life[m],90):.1f} max {life[m].max():.1f}")
hw = rec[:, 2]; cu = (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
slow = fr & (life > 70)
print("slow first-round wgs:", slow.sum(), "distinct (xcc, cu):", len(set(zip(xcc[slow].tolist(), cu[slow].tolist()))), "of", len(set(zip(xcc[fr].tolist(), cu[fr].tolist()))))
per = collections.Counter(zip(xcc[slow].tolist(), cu[slow].tolist()))
print("slow wgs per CU histogram:", sorted(collections.Counter(per.values()).items()))
print("slow wgs: block index deciles", [int(x) for x in np.percentile(blk[slow], range(0, 101, 20))], "frame-local order (blk>>3)%4:", collections.Counter(q[slow].tolist()))
for lo, hi in ((0, 2048), (2048, 3584), (3584, 4096), (4096, 4160), (4160, 5000), (5000, 8192), (8192, 12288), (12288, 1 << 20)):
    m = fr & (gc > lo) & (gc <= hi)
    if m.sum():
        print(f"first round, {lo} < candidates <= {hi}: n {m.sum()} life median {np.median(life[m]):.1f} min {life[m].min():.1f} max {life[m].max():.1f}")
se = (hw >> 13) & 7
pos = (blk >> 3) % 4   # the workgroup's position among its frame's four = its launch order in the XCD, mod 4
print("shader engine (HW_ID bits 13-15) by launch order in the XCD mod 4, first round:")
for k in range(4):
    print("  order", k, "->", sorted(collections.Counter(se[fr & (pos == k)].tolist()).items()))
