"""GPU box: how many (writer, raster band) code lists overflow at a given capacity (BEV_CODE_CAP) on the synthetic layouts —
what kCodeListCap (bev_internal.h) was chosen from.  usage: python3 scripts/list_fill.py"""
import sys, os
sys.path.insert(0,'point-cloud-preprocessing-tools_amd'); sys.path.insert(0,'tests')
import numpy as np, bev_amd
from bev_amd import synth
for sensor, maker in [("OS1_64", lambda p,i: synth.firing_order(p,i)), ("HDL_32E", lambda p,i: synth.sweep(p,i)), ("HDL_64E", lambda p,i: synth.sweep(p,i)),
                      ("HDL_32E", lambda p,i: synth.firing_order(p,i)), ("HDL_64E", lambda p,i: synth.firing_order(p,i)), ("OS1_64", lambda p,i: synth.sweep(p,i))]:
    p = bev_amd.params_for_sensor(sensor)
    frames=[maker(p,i) for i in range(6)]
    res=[]
    for cap in (1024, 1536, 2048, 2560, 3072, 4096):
        os.environ["BEV_CODE_CAP"]=str(cap)
        ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
        ctx.process_batch(frames)
        res.append((cap, int(ctx.code_overflow(0,6).sum())))
        ctx.close()
    print(sensor, len(frames[0]), res, flush=True)
