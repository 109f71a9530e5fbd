#!/usr/bin/env python3
"""Frames of EVERY layout mixed at random through the device-resident, pipelined path — small sub-batches, 1-3 workspace
sets, back-to-back asynchronous calls — every frame against the oracle.  What it is after: the walks of modes a workspace
set has not seen lately are not launched (a hint k_verdict leaves in mapped host memory, read without waiting), frames
whose walk was skipped are redone the general way; runs of one layout alternate with mixtures so that modes come and go.
usage (GPU box): python3 scripts/mixed_soak.py [seed] [rounds] [sensor]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd")); sys.path.insert(0, str(REPO / "tests"))
import numpy as np, torch
import bev_amd, oracle_lib as orc
from bev_amd import synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sensor = sys.argv[3] if len(sys.argv) > 3 else "HDL_32E"
p = bev_amd.params_for_sensor(sensor)
sp = orc.sensor_from_params(p)
S, M, L = p.slots, p.mat_size, p.n_layers
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)


def make(kind, fid):
    if kind == 0: return synth.sweep(p, fid, keep=0.97, n_dup=int(300 * S / 33792))
    if kind == 1: return synth.structured(p, fid, 0.95, kitti_intensity=bool(fid % 2))
    if kind == 2: return synth.firing_order(p, fid)
    if kind == 6: return synth.firing_real(p, fid, noret=(0.0, 0.03, 0.3)[fid % 3])   # a real MulRan sweep: phase, direction, stagger, no-returns
    if kind == 3: return synth.adversarial(p, S // 2 + fid % 1000, fid, False)
    if kind == 4: return synth.sweep(p, fid, keep=1.0, n_dup=0)            # structured and sorted at once
    f = synth.structured(p, fid, 1.0)                                      # a structured cloud with ONE hidden empty record: wrong guess, redone
    k = 1000 + fid % 5000
    while k % 63 in (0, 1):
        k += 1
    f[k] = np.zeros(1, bev_amd.POINT_DTYPE)[0]
    return f


bad_total, modes = 0, {}
t00 = time.time()
for rnd in range(rounds):
    n = int(rng.integers(60, 400))
    sub = int(rng.choice([5, 16, 33, 100]))
    lanes = int(rng.choice([1, 2, 3]))
    # runs of one layout, then mixtures
    kinds = []
    while len(kinds) < n:
        if rng.random() < 0.5:
            kinds += [int(rng.integers(0, 7))] * int(rng.integers(1, 3 * sub))
        else:
            kinds += [int(k) for k in rng.integers(0, 7, int(rng.integers(1, 2 * sub)))]
    kinds = kinds[:n]
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda a: make(a[1], 100000 * rnd + a[0]), enumerate(kinds)))
    os.environ["BEV_LANES"] = str(lanes)
    ctx = bev_amd.BevContext(p, device=0, max_batch=sub, max_points=max(len(f) for f in frames))
    os.environ.pop("BEV_LANES")
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(f) for f in frames])
    d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
    outs = [torch.zeros(n * k, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
    for _ in range(int(rng.integers(1, 4))):   # back-to-back asynchronous calls over the same buffers
        ctx.process_device(n, d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
    ctx.synchronize()
    last = min(sub, n - (n - 1) // sub * sub)
    for m in ctx.frame_info(0, last)[:, 1]:
        modes[int(m)] = modes.get(int(m), 0) + 1
    ords, multis, singles = (o.cpu().numpy() for o in outs)
    ctx.close()

    def check(i):
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes() and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())
    with ThreadPoolExecutor(16) as ex:
        bad = [i for i, ok in enumerate(ex.map(check, range(n))) if not ok]
    bad_total += len(bad)
    print(f"round {rnd}: {n} frames, sub-batch {sub}, {lanes} sets: {len(bad)} differ {[(i, kinds[i]) for i in bad[:5]]}", flush=True)
print(f"mixed soak ({sensor}): seed {seed}, {rounds} rounds, {bad_total} mismatches, modes of the last sub-batches {modes}, {time.time() - t00:.0f} s")
sys.exit(1 if bad_total else 0)
