"""PCIe-inclusive rate of bev_process_batch (host buffers in, host buffers out) — not the benchmark value."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-cloud-preprocessing-tools_amd"))
import numpy as np
import bev_amd
from bev_amd import synth, POINT_DTYPE

p = bev_amd.params_for_sensor("HDL_64E")
nf = int(os.environ.get("NF", "256"))
batch = int(os.environ.get("BATCH", "64"))
frames = [synth.sweep(p, s) for s in range(8)]
clouds = [frames[i % 8].copy() for i in range(nf)]
ctx = bev_amd.BevContext(p, max_batch=batch, max_points=140000)
S, L, M = ctx.S, ctx.L, ctx.M
pinned = os.environ.get("PINNED", "0") == "1"
if pinned:
    alloc = bev_amd.host_alloc
    src = [alloc(len(f), POINT_DTYPE) for f in clouds]
    for d, s_ in zip(src, clouds): d[:] = s_
    clouds = src
else:
    alloc = lambda shape, dt: np.zeros(shape, dt)
ordered = alloc((nf, S), POINT_DTYPE); multi = alloc((nf, L, M, M), np.uint8); single = alloc((nf, M, M), np.uint8)
VP = C.c_void_p * nf
pts = VP(*[c.ctypes.data for c in clouds]); npts = (C.c_uint32 * nf)(*[len(c) for c in clouds])
o = VP(*[ordered[i].ctypes.data for i in range(nf)]); m = VP(*[multi[i].ctypes.data for i in range(nf)])
s = VP(*[single[i].ctypes.data for i in range(nf)])
mb = sum(c.nbytes for c in clouds) + ordered.nbytes + multi.nbytes + single.nbytes
for rep in range(4):
    t = time.time()
    rc = ctx.lib.bev_process_batch(ctx._h, nf, pts, npts, o, m, s, None)
    dt = time.time() - t
    assert rc == 0
    print(f"rep {rep}: {nf / dt:.0f} frames/s, {mb / dt / 1e9:.1f} GB/s over PCIe (both directions), batch {batch}, {'pinned' if pinned else 'pageable'} host buffers")
