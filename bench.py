#!/usr/bin/env python3
"""bench.py — BEV frames/s of the batch_multi_bev_gen hot path on MI355X.

Workload (BASELINE.json configs[1]; configs[3] is the same per GPU): 1000
synthetic KITTI-like HDL_64E clouds (~135.6k input points each into 133,312
slots) per GPU, resident in HBM; one "step" = one pass of the whole hot path
(order -> ground segmentation -> multi + single BEV) over those 1000 frames,
outputs left in HBM.  The timed region runs the library's default mode — fused launches
(k_stage) over two streams —, the roofline of the dominant kernel comes from a pass of the
same steps with one launch per kernel (bev_set_lanes(ctx, 1)).  N > 1: one process per GPU (torch.distributed, backend
nccl = RCCL), frames sharded contiguously, the only collective on the path is
the broadcast of the frame-range table; weak scaling (1000 frames per GPU).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd"))
sys.path.insert(0, str(REPO / "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# What byte-moving kernels reach on the pool's boxes (scripts/microbench/copyshapes.hip, tileshapes.hip; raw output under
# profiles/r03_copy_ceiling_box1.txt, r03_tile_shapes_box1.txt): a no-loop 16-B-per-thread copy, and the column walk's
# geometry (a 64-row loop over 8 KiB row pieces, whole-line stores) as a pure copy.
COPY_CEILING_GBPS = 6610.0
WALK_SHAPE_CEILING_GBPS = 5010.0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps (a step = the whole path over the frames; 20 x 2.5 ms: the bench is mostly set-up and the CPU baseline)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU (BASELINE config: 1000)")
    ap.add_argument("--sensor", default=None)
    ap.add_argument("--workload", default="hdl64_sweep", choices=["hdl64_sweep", "os1_firing", "oxford_concat", "hdl64_structured", "os1_firing_real", "mixed", "hdl64_shuffled"],
                    help="hdl64_sweep = BASELINE configs[1]/[3] (default, the graded metric); os1_firing = configs[2] "
                         "(MulRan-style unordered OS1_64); oxford_concat = configs[4] (HDL_32E, ~2M points per frame); "
                         "hdl64_structured = the same sweeps in the layout the reference's KITTI selector writes "
                         "(KittiPointCloudSelect.cpp:206-207,240: S records, dropped returns as all-zero records; not a "
                         "BASELINE config, named in config.workload); hdl64_shuffled = configs[1]'s frames with their points in a "
                         "random order (what getOrderedCloud's contract allows, BatchMultiBevGen.cpp:102-116: the general path — "
                         "order scan + gather walk —, the route of any layout the probe does not recognise and of every frame that "
                         "fails its checks; not the graded metric)")
    ap.add_argument("--sub-batch", type=int, default=int(os.environ.get("BEV_SUB_BATCH", "500")),
                    help="frames per sub-batch (= per fused launch: a sub-batch's walk beside the later stages of earlier ones; 500: two per "
                         "1000-frame step; 125 / 250 / 334 / 1000 measured 2-3 % behind 500, profiles/r06_experiments.txt)")
    ap.add_argument("--n-dup", type=int, default=int(os.environ.get("BEV_BENCH_NDUP", "5000")),
                    help="duplicates appended to every hdl64_sweep frame (BASELINE config 2: 5000; anything else is a "
                         "developer experiment and is named in config.workload)")
    ap.add_argument("--layout-hint", default="none", choices=["none", "structured", "firing"],
                    help="bev_set_layout_hint: the caller says how its clouds are laid out and k_probe does not look (verified by the "
                         "walk like any guess); named in config.workload")
    ap.add_argument("--cpu-sample", type=int, default=400, help="frames timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-build", action="store_true",
                    help="do not run __graft_entry__.build() first (profiling scripts build once, then run "
                         "`rocprofv3 ... -- python3 bench.py --no-build`: no child process under the profiler)")
    args = ap.parse_args()

    import numpy as np
    import torch  # before the HIP library: one HIP runtime per process (bev_amd.load_lib)
    import torch.distributed as dist

    import __graft_entry__ as ge

    # ---- everything up to here has made no GPU call (torch.cuda.device_count() does not initialise the device) ----
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: start N ranks (one process per GPU) and relay rank 0's JSON line.
        # This parent never touches the GPU.
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this machine shows {n_dev}", file=sys.stderr)
            return 2
        if not args.no_build:
            ge.build()
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        child_args = [a for a in sys.argv[1:]]
        if "--no-build" not in child_args:
            child_args.append("--no-build")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *child_args]
        return subprocess.run(cmd).returncode

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not args.no_build:
        # before the first GPU call of this process (make / hipcc children must not be forked from a process that holds
        # the GPU); ranks of one node serialise on a file lock, the build is a no-op when everything is up to date
        import fcntl
        with open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "bev_bench_build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            ge.build()
    import bev_amd
    from bev_amd import shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to fall back to)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run even a single rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    lib_missing = not bev_amd.LIB_PATH.exists()
    if lib_missing:
        raise SystemExit(f"{bev_amd.LIB_PATH} missing")

    default_sensor = {"hdl64_sweep": "HDL_64E", "os1_firing": "OS1_64", "oxford_concat": "HDL_32E",
                      "hdl64_structured": "HDL_64E", "os1_firing_real": "OS1_64", "mixed": "HDL_64E", "hdl64_shuffled": "HDL_64E"}[args.workload]
    args.sensor = args.sensor or default_sensor
    p = bev_amd.params_for_sensor(args.sensor)
    S, M, L = p.slots, p.mat_size, p.n_layers
    F = args.frames

    # ---- frame-range table: rank 0 decides, RCCL broadcast (the path's only collective)
    table = shard.broadcast_ranges(F * world, rank, world, device=dev, force_collective=use_dist)
    first, count = int(table[rank, 0]), int(table[rank, 1])
    assert count == F

    # ---- synthetic frames, generated on the host cores, then made resident in HBM
    n_dup = args.n_dup
    n_sweeps = 60
    cap = {"hdl64_sweep": S + n_dup, "os1_firing": S, "oxford_concat": S * n_sweeps, "hdl64_structured": S,
           "os1_firing_real": S, "mixed": S + n_dup, "hdl64_shuffled": S + n_dup}[args.workload]
    t_gen = time.time()
    host = np.empty((count, cap), dtype=bev_amd.POINT_DTYPE)
    counts = np.zeros(count, dtype=np.int64)

    def gen(i):
        if args.workload == "hdl64_sweep":
            counts[i] = len(synth.sweep(p, first + i, keep=0.98, n_dup=n_dup, out=host[i]))
        elif args.workload == "hdl64_shuffled":
            n_i = len(synth.sweep(p, first + i, keep=0.98, n_dup=n_dup, out=host[i]))
            host[i, :n_i] = host[i, :n_i][np.random.default_rng(0x5EED0000 + first + i).permutation(n_i)]
            counts[i] = n_i
        elif args.workload == "os1_firing":
            pts = synth.firing_order(p, first + i)
            host[i, :len(pts)] = pts
            counts[i] = len(pts)
        elif args.workload == "hdl64_structured":
            host[i] = synth.structured(p, first + i, keep=0.98)
            counts[i] = S
        elif args.workload == "os1_firing_real":
            # what mulran_point_cloud_select writes for real sweeps (MulranPointCloudSelect.cpp:112-130): 3 % no-return
            # records (column 0 of their row), start azimuth and direction drawn per frame, staggered laser columns
            # (developer knobs for traffic diagnosis: BEV_FR_NORET / BEV_FR_STAGGER / BEV_FR_PHASE / BEV_FR_DIR override the draw)
            host[i] = synth.firing_real(p, first + i, noret=float(os.environ.get("BEV_FR_NORET", "0.03")),
                                        stagger=float(os.environ.get("BEV_FR_STAGGER", "1.0")),
                                        phase=int(os.environ["BEV_FR_PHASE"]) if "BEV_FR_PHASE" in os.environ else None,
                                        direction=int(os.environ["BEV_FR_DIR"]) if "BEV_FR_DIR" in os.environ else None)
            counts[i] = S
        elif args.workload == "mixed":
            # the layouts of the reference's producers alternating in groups of 32 frames (the CLI's batch): sorted sweeps
            # with appended duplicates, structured clouds, firing order
            kind = ((first + i) // 32) % 3
            if kind == 0:
                counts[i] = len(synth.sweep(p, first + i, keep=0.98, n_dup=n_dup, out=host[i]))
            elif kind == 1:
                host[i, :S] = synth.structured(p, first + i, keep=0.98)
                counts[i] = S
            else:
                host[i, :S] = synth.firing_real(p, first + i, noret=0.03)
                counts[i] = S
        else:
            pts = synth.concat(p, first + i, n_sweeps=n_sweeps)
            host[i, :len(pts)] = pts
            counts[i] = len(pts)

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        list(ex.map(gen, range(count)))
    offsets = np.zeros(count + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    total_pts = int(offsets[-1])
    d_in = torch.empty(total_pts * 32, dtype=torch.uint8, device=dev)
    for i in range(count):
        a, b = int(offsets[i]) * 32, int(offsets[i + 1]) * 32
        d_in[a:b].copy_(torch.from_numpy(host[i, : counts[i]].view(np.uint8).reshape(-1)))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    d_ordered = torch.empty(count * S * 32, dtype=torch.uint8, device=dev)
    d_multi = torch.empty(count * L * M * M, dtype=torch.uint8, device=dev)
    d_single = torch.empty(count * M * M, dtype=torch.uint8, device=dev)

    ctx = bev_amd.BevContext(p, device=local_rank, max_batch=args.sub_batch, max_points=int(counts.max()))

    if args.layout_hint != "none":
        ctx.set_layout_hint(bev_amd.LAYOUT_STRUCTURED if args.layout_hint == "structured" else bev_amd.LAYOUT_FIRING_ORDER)

    def step():
        ctx.process_device(count, d_in.data_ptr(), offsets, d_ordered.data_ptr(), d_multi.data_ptr(),
                           d_single.data_ptr())

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    elapsed = shard.max_over_ranks(elapsed, world, device=dev, force_collective=use_dist)
    # the same steps once more, each one fenced by itself: the median step next to the mean of the timed region
    # (SURVEY.md 8(d) asks for the median; a fence per step costs the pipeline its overlap across steps)
    per_step = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        fence()
        per_step.append(time.perf_counter() - ts)
    per_step.sort()
    median_step = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    total_frames = shard.sum_over_ranks(float(count * args.steps), world, device=dev, force_collective=use_dist)
    # what every rank measured by itself (its shard of the table, its own wall time of the timed region, its device): the
    # line shows that the collectives ran over `ranks_seen` ranks without anyone reading logs
    ranks_seen = dist.get_world_size() if use_dist else 1
    per_rank = shard.gather_per_rank([rank, local_rank, first, count, local_elapsed, count * args.steps / local_elapsed],
                                     world, device=dev, force_collective=use_dist)

    # ---- profile passes (after the timed region, which carries no events): (1) the same steps again, still pipelined,
    # every launch bracketed by HIP events on its own stream: what each kernel costs UNDER OVERLAP, i.e. in the
    # configuration the headline number comes from; (2) the same steps with ONE lane, launches back to back: a
    # launch's duration is then its own, which is what the roofline of the dominant kernel is computed from.
    stats, stats_pipe = [], []
    if not args.no_profile:
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(args.steps):
            step()
        fence()
        stats_pipe = ctx.profile_get()
        ctx.profile_enable(False)
        ctx.set_lanes(1)
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(args.steps):
            step()
        fence()
        stats = ctx.profile_get()
        ctx.profile_enable(False)

    n_last = count - ((count - 1) // args.sub_batch) * args.sub_batch
    routes = {str(k): int(v) for k, v in zip(*np.unique(ctx.frame_info(0, n_last)[:, 1], return_counts=True))}
    # ---- per-kernel HIP-event durations of the roofline pass (this rank)
    mean_pts = total_pts / count
    b_frame = bev_amd.algorithmic_bytes_per_frame(p, mean_pts)  # 32P + 32S + L*M*M + M*M
    # which part of B_frame each kernel is the one to move (DESIGN.md "Kernels"): the input is counted ONCE, for the kernel
    # whose read of it cannot be avoided — the column walk when P ~ S (it gathers every point it writes), the order scan
    # when P >> S (oxford_concat: 2 M points into 33,792 slots; the walk then only gathers the S winners)
    if args.workload in ("oxford_concat", "hdl64_shuffled"):  # (shuffled: scattered atomics make the scan the longest kernel; it is the one that streams the input)
        own_bytes = {"k_order_scan": 32.0 * mean_pts, "k_walk": 32.0 * S, "k_walk_general": 32.0 * S,
                     "k_bev_raster": float(L * M * M + M * M)}
    else:  # k_walk: frames read in place; k_walk_general: frames that go through the winner table (only one of the two moves a frame)
        own_bytes = {"k_walk": 32.0 * mean_pts + 32.0 * S, "k_walk_general": 32.0 * mean_pts + 32.0 * S,
                     "k_walk_structured": 32.0 * mean_pts + 32.0 * S, "k_walk_colmajor": 32.0 * mean_pts + 32.0 * S,
                     "k_walk_colmajor_gen": 32.0 * mean_pts + 32.0 * S,
                     "k_bev_raster": float(L * M * M + M * M)}
    roofline = None
    kernels = []
    kernels_pipelined = [{"name": s["name"], "launches": s["launches"], "avg_launch_ms": s["total_ms"] / s["launches"],
                          "total_ms": s["total_ms"]} for s in stats_pipe]
    if stats:
        tot_ms = sum(s["total_ms"] for s in stats)
        for s in stats:
            per_launch_frames = s["frames"] / s["launches"]
            kernels.append({
                "name": s["name"], "launches": s["launches"],
                "avg_launch_ms": s["total_ms"] / s["launches"],
                "share": s["total_ms"] / tot_ms,
                "algorithmic_bytes_per_launch": own_bytes.get(s["name"], 0.0) * per_launch_frames,
            })
        dom = max(stats, key=lambda s: s["total_ms"])
        per_launch_frames = dom["frames"] / dom["launches"]
        if args.workload == "mixed":  # a walk moves only the frames of its route: their share of the last sub-batch stands for every launch
            route_of = {"k_walk": ("1",), "k_walk_structured": ("3",), "k_walk_colmajor": ("4",), "k_walk_colmajor_gen": ("5",), "k_walk_general": ("0", "2")}
            share = sum(routes.get(r, 0) for r in route_of.get(dom["name"], ())) / max(1, sum(routes.values()))
            if dom["name"] in route_of:
                per_launch_frames *= share
        avg_ms = dom["total_ms"] / dom["launches"]
        dom_bytes = own_bytes.get(dom["name"], 0.0) * per_launch_frames
        achieved = dom_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, gfx950 FETCH correction applied; see profiles/): per frame, scaled to this launch size
        # PMC counters cannot be read from inside the process: they come from the committed rocprofv3 --pmc passes of
        # THIS command line (scripts/profile_round.sh; 1000 frames, sub-batch 500), per frame, scaled to this launch.
        traffic, traffic_src, traffic_total = None, None, None
        # which instantiation of the walk a profile id is (rocprofv3 names kernels by their template arguments)
        pmc_prefix = {"k_walk": "k_walk<2,", "k_walk_general": "k_walk<0,", "k_walk_structured": "k_walk<3,",
                      "k_walk_colmajor": "k_walk<4,", "k_walk_colmajor_gen": "k_walk<5,"}.get(dom["name"], dom["name"])
        tag = "" if args.workload == "hdl64_sweep" else args.workload + "_"
        rounds = ("r06", "r05", "r04") if args.workload != "hdl64_sweep" else ("r06", "r05", "r04", "r03", "r02")
        for name in (f"{r}_{tag}pmc_traffic.json" for r in rounds):
            pmc_file = REPO / "profiles" / name
            if not pmc_file.exists():
                continue
            pmc = json.loads(pmc_file.read_text())
            for kname, kv in pmc["kernels"].items():
                if kname.startswith(pmc_prefix) and kv.get("hbm_bytes_per_frame", 0) > 1e5:  # (the instantiation that moved the frames, not an empty launch)
                    traffic = kv["hbm_bytes_per_frame"] * per_launch_frames
                    traffic_src = f"profiles/{name}: {pmc.get('source', 'rocprofv3 --pmc')}"
            traffic_total = pmc.get("hbm_bytes_per_frame_all_kernels")
            break
        # whole hot path against the wall clock of the timed region (this rank): B_frame * frames / time
        pipe_achieved = b_frame * (count * args.steps) / elapsed / 1e9
        frames_per_s = count * args.steps / elapsed
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
            "copy_ceiling_gbps": COPY_CEILING_GBPS, "walk_shape_ceiling_gbps": WALK_SHAPE_CEILING_GBPS,
            "ceiling_source": "profiles/r03_copy_ceiling_box1.txt, profiles/r03_tile_shapes_box1.txt (no-loop copy; the walk's geometry as a pure copy)",
            "kernel": dom["name"], "frames_per_launch": per_launch_frames,
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": avg_ms,
            "note": "kernel durations from a one-lane pass of the same steps right after the timed region (back-to-back launches); "
                    "the timed region itself runs FUSED launches (k_stage: a sub-batch's walk beside the later stages of earlier sub-batches, two streams)",
            "pipeline": {"bytes_per_frame": b_frame, "achieved": pipe_achieved, "frac": pipe_achieved / HBM_PEAK_GBPS,
                         "hbm_traffic_per_frame_all_kernels": traffic_total,
                         "real_traffic_gbps": (traffic_total * frames_per_s / 1e9) if traffic_total else None,
                         "definition": "algorithmic bytes of the whole hot path / wall time of the timed region, this GPU"},
        }

    # ---- CPU baseline: the oracle (a port of the reference algorithm), 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle_lib as orc

        sp = orc.sensor_from_params(p)
        n_cpu = min(args.cpu_sample, count)
        o_ord = np.empty(S, bev_amd.POINT_DTYPE)
        o_multi = np.empty((L, M, M), np.uint8)
        o_single = np.empty((M, M), np.uint8)
        import ctypes as C
        lib = orc.lib()
        tc = time.perf_counter()
        for i in range(n_cpu):
            fr = host[i, : counts[i]]
            lib.oracle_process_frame(C.byref(sp), fr.ctypes.data, len(fr), o_ord.ctypes.data, None,
                                     o_multi.ctypes.data, o_single.ctypes.data)
        tc = time.perf_counter() - tc
        # the last oracle frame doubles as a live parity check of the benchmarked run
        i = n_cpu - 1
        got = d_ordered[i * S * 32:(i + 1) * S * 32].cpu().numpy().tobytes()
        parity = (got == o_ord.tobytes()
                  and d_multi[i * L * M * M:(i + 1) * L * M * M].cpu().numpy().tobytes() == o_multi.tobytes()
                  and d_single[i * M * M:(i + 1) * M * M].cpu().numpy().tobytes() == o_single.tobytes())
        # the same oracle frame-parallel on the host's cores (the reference is single-threaded: context only).  ctypes
        # releases the GIL, every worker has its own output buffers; bounded to the same frames as above.
        n_thr = max(1, min(os.cpu_count() or 1, 16))  # a 1-GPU box has a CPU share of 16
        bufs = [(np.empty(S, bev_amd.POINT_DTYPE), np.empty((L, M, M), np.uint8), np.empty((M, M), np.uint8)) for _ in range(n_thr)]

        def cpu_worker(w):
            o, m, s_ = bufs[w]
            for i in range(w, n_cpu, n_thr):
                fr = host[i, : counts[i]]
                lib.oracle_process_frame(C.byref(sp), fr.ctypes.data, len(fr), o.ctypes.data, None, m.ctypes.data, s_.ctypes.data)

        tp = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_thr) as ex:
            list(ex.map(cpu_worker, range(n_thr)))
        tp = time.perf_counter() - tp
        # the reference's own timed region (BatchMultiBevGen.cpp:732-752) also writes the .bin and the .csv of every
        # frame (PNG encodes excluded here): the same oracle calls plus those two files, on a smaller sample
        import tempfile
        n_io = min(100, n_cpu)
        with tempfile.TemporaryDirectory(prefix="bev_cpu_") as td:
            tio = time.perf_counter()
            for i in range(n_io):
                fr = host[i, : counts[i]]
                lib.oracle_process_frame(C.byref(sp), fr.ctypes.data, len(fr), o_ord.ctypes.data, None,
                                         o_multi.ctypes.data, o_single.ctypes.data)
                rc_io = lib.oracle_save_bin_csv(o_multi.ctypes.data, o_single.ctypes.data, M, L,
                                                os.path.join(td, f"{i:06d}.bin").encode(), os.path.join(td, f"{i:06d}.csv").encode())
                assert rc_io == 0
            tio = time.perf_counter() - tio
        cpu_model = ""
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            pass
        cpu = {"value": n_cpu / tc, "unit": "frames/s", "cores": 1, "kind": "port",
               "frame_parallel": {"value": n_cpu / tp, "unit": "frames/s", "cores": n_thr},
               "timed_region": {"value": n_io / tio, "unit": "frames/s", "cores": 1, "ms_per_frame": tio / n_io * 1e3,
                                "sample": f"first {n_io} frames: oracle_process_frame + the .bin (1,204,224 B) and .csv (250,656 B) "
                                          "writes the reference's timer covers (BatchMultiBevGen.cpp:732-752), PNG encodes excluded, "
                                          "files on the box's temp dir"},
               "cpu_model": cpu_model,
               "sample": f"first {n_cpu} of the {count} frames, oracle_process_frame (order+ground+both rasters, outputs to memory), gcc -O3 no -march",
               "ms_per_frame": tc / n_cpu * 1e3, "host_cpus": os.cpu_count(),
               "gpu_output_matches_oracle_on_sampled_frame": bool(parity)}

    if rank == 0:
        out = {
            "metric": "BEV frames/sec (131k-pt HDL-64E cloud)" if args.workload == "hdl64_sweep"
                      else f"BEV frames/sec ({args.workload})",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,  # dist.get_world_size() after init_process_group (backend nccl = RCCL); 1 without torch.distributed
            "frame_range_table": [[int(a), int(b)] for a, b in table],  # what rank 0 broadcast: [first, count] per rank
            "per_rank": [{"rank": int(r[0]), "local_rank": int(r[1]), "first_frame": int(r[2]), "frames": int(r[3]),
                          "timed_region_s": float(r[4]), "frames_per_s": float(r[5])} for r in per_rank],  # all_gather over the ranks
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median_fenced": median_step * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{F} synthetic {args.workload} {args.sensor} clouds per GPU (mean {mean_pts:.0f} input pts, "
                                   f"{S} slots), single+multi BEV, device-resident" +
                                   (f" [developer run: {n_dup} appended duplicates instead of 5000]" if args.workload == "hdl64_sweep" and n_dup != 5000 else "") +
                                   (f" [bev_set_layout_hint: {args.layout_hint}]" if args.layout_hint != "none" else ""),
                       "frames_per_gpu": F, "frames_per_step": F * world, "sub_batch": args.sub_batch, "sensor": args.sensor,
                       "algorithmic_bytes_per_frame": b_frame, "parallelism": f"frames x{world}"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "kernels_pipelined": kernels_pipelined,
            "gen_seconds": t_gen,
            # how the frames of the last sub-batch reached their slots (bev_debug_get_frame_info): 0 general, 1 sorted prefix read
            # in place, 2 read in place, failed its checks and redone, 3 structured cloud, 4 firing order (the plain sweep), 5 firing order (any phase,
            # direction, stagger, no-return records)
            "routes_last_sub_batch": routes,
        }
        print(json.dumps(out))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
