"""Soak: many different BASELINE-config frames through the pipelined device-resident path, EVERY frame compared with the
oracle.  usage (GPU box): python scripts/soak.py [rounds] [frames_per_round] [hdl64_sweep | os1_firing | os1_firing_real | hdl64_firing_real | hdl32_sweep | hdl64_adversarial | hdl64_structured | hdl64_gappy] [sub_batch]
(hdl64_gappy: sorted sweeps with BURSTS of dropped returns — runs of 1 to 400 slots at random places, next to row starts and strip
boundaries too —, a tenth of the returns invalid, appended points: what shifts the in-place source's windows off their spans)
(sub_batch defaults to 500, bench.py's launch size)"""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd")); sys.path.insert(0, str(REPO / "tests"))
import numpy as np, torch
import bev_amd, oracle_lib as orc
from bev_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
workload = sys.argv[3] if len(sys.argv) > 3 else "hdl64_sweep"
sub_batch = int(sys.argv[4]) if len(sys.argv) > 4 else 500
p = bev_amd.params_for_sensor({"hdl64_sweep": "HDL_64E", "os1_firing": "OS1_64", "hdl32_sweep": "HDL_32E", "hdl64_adversarial": "HDL_64E",
                               "hdl64_structured": "HDL_64E", "hdl64_gappy": "HDL_64E", "os1_firing_real": "OS1_64", "hdl64_firing_real": "HDL_64E"}[workload])
sp = orc.sensor_from_params(p)


def make_frame(rnd, f):
    fid = 100000 + rnd * n + f
    if workload == "os1_firing":
        return synth.firing_order(p, fid)
    if workload in ("os1_firing_real", "hdl64_firing_real"):
        # real MulRan sweeps (round 5): no-return shares from none to a third, every start azimuth and direction, staggers from
        # none to full; a tenth of the frames with invalid returns (intensity -1: phase A's fallbacks cross the row ends)
        rng = np.random.default_rng(fid)
        fr = synth.firing_real(p, fid, noret=float(rng.choice([0.0, 0.002, 0.03, 0.03, 0.1, 0.33])),
                               stagger=float(rng.choice([0.0, 0.5, 1.0, 1.0])))
        if f % 10 == 3:
            fr["intensity"][rng.random(len(fr)) < float(rng.choice([0.02, 0.2]))] = -1.0
        return fr
    if workload == "hdl64_gappy":
        rng = np.random.default_rng(fid)
        base = synth.sweep(p, fid, keep=1.0 - 0.01 * (f % 4), n_dup=(0, 200, 3000)[f % 3])
        keep = np.ones(len(base), bool)
        n_sorted = len(base) - (0, 200, 3000)[f % 3]
        H = p.horizon_scan
        for _ in range(int(rng.integers(0, 60))):
            ln = int(min(400, rng.geometric(0.02)))
            where = int(rng.integers(0, 4))
            row = int(rng.integers(0, p.n_scan))
            if where == 0:   start = row * H + int(rng.integers(0, 8))                              # just after a row start
            elif where == 1: start = row * H + 236 * int(rng.integers(1, 9)) - int(rng.integers(0, ln + 4))   # around a strip boundary
            elif where == 2: start = row * H + H - ln - int(rng.integers(0, 6))                       # before the row's end
            else:            start = int(rng.integers(0, n_sorted))
            slot = base["row"][:n_sorted].astype(np.int64) * H + base["col"][:n_sorted]
            keep[:n_sorted] &= ~((slot >= start) & (slot < start + ln))
        out = np.ascontiguousarray(base[keep])
        out["intensity"][rng.random(len(out)) < 0.1] = -1.0
        return out
    if workload == "hdl64_structured":
        return synth.structured(p, fid, keep=0.98 - 0.3 * (rnd % 3), kitti_intensity=bool(rnd % 2))
    if workload == "hdl64_adversarial":
        return synth.adversarial(p, 60000 + 1000 * (f % 70), fid, nonfinite=bool(f % 2))
    return synth.sweep(p, fid, keep=0.98 - 0.02 * (rnd % 3), n_dup=5000 + 500 * (rnd % 4))


S, M, L = p.slots, p.mat_size, p.n_layers
dev = torch.device("cuda:0")
ctx = bev_amd.BevContext(p, device=0, max_batch=sub_batch, max_points=max(S + 8000, 140000))
print(f"soak: {workload}, {rounds} x {n} frames, sub-batch {sub_batch}", flush=True)
bad_total = 0
for rnd in range(rounds):
    t0 = time.time()
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda f: make_frame(rnd, f), range(n)))
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(f) for f in frames])
    d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
    d_ord = torch.zeros(n * S * 32, dtype=torch.uint8, device=dev)
    d_multi = torch.zeros(n * L * M * M, dtype=torch.uint8, device=dev); d_single = torch.zeros(n * M * M, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # torch's fills and copies run on ITS stream; the library's streams do not wait for it
    for _ in range(2):
        ctx.process_device(n, d_in.data_ptr(), offs, d_ord.data_ptr(), d_multi.data_ptr(), d_single.data_ptr())
    ctx.synchronize()
    ords, multis, singles = d_ord.cpu().numpy(), d_multi.cpu().numpy(), d_single.cpu().numpy()
    def check(i):
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes() and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())
    with ThreadPoolExecutor(16) as ex:
        bad = [i for i, ok in enumerate(ex.map(check, range(n))) if not ok]
    bad_total += len(bad)
    print(f"round {rnd}: {n} frames, {len(bad)} differ from the oracle {bad[:5]}, {time.time() - t0:.1f} s", flush=True)
    del d_in, d_ord, d_multi, d_single
print("TOTAL frames", rounds * n, "mismatches", bad_total)
sys.exit(1 if bad_total else 0)
