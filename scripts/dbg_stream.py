import os, sys
os.environ["BEV_STREAM"]="1"
sys.path.insert(0,"/root/repo/tests"); sys.path.insert(0,"/root/repo/point-cloud-preprocessing-tools_amd")
import numpy as np, bev_amd
from bev_amd import synth
p = bev_amd.params_for_sensor("HDL_32E")
print(p.n_scan, p.horizon_scan)
frames = [synth.sweep(p, 1), synth.firing_order(p, 2), synth.sweep(p, 3, keep=0.5, n_dup=9000), np.empty(0, bev_amd.POINT_DTYPE), synth.sweep(p, 5)]
ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
ctx.process_batch(frames, want_ground_mat=True)
print([len(f) for f in frames]); print(ctx.frame_info(0, 5))
