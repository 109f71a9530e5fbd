"""Turn a scripts/profile_round.sh result directory (gpurun_out/<tag>/) into the files kept under profiles/:
   <prefix>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary (one lane)
   <prefix>_bench_under_rocprof.json  bench.py line of that profiled run
   <prefix>_bench.json                bench.py line of the unprofiled default run (with the CPU baseline)
   <prefix>_pmc_traffic.json          per-kernel HBM bytes from the PMC passes (gfx950 corrections applied)
   <prefix>_<workload>_bench.json / _kernel_stats.csv   the same for BASELINE configs 3 and 5 when present
usage: python scripts/make_profiles.py gpurun_out/<tag> profiles/<prefix> [frame passes of a PMC run, default 2000]"""
import collections, csv, glob, json, os, shutil, sys

src, prefix = sys.argv[1], sys.argv[2]
pmc_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3000  # bench.py --frames 1000 --steps 1 --warmup 1: warm-up, timed step, fenced step


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], prefix + "_kernel_stats.csv")
for name, out in (("bench_under_rocprof.log", "_bench_under_rocprof.json"), ("bench.log", "_bench.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        json.dump(last_json_line(p), open(prefix + out, "w"), indent=1)

for wl in ("os1_firing", "oxford_concat", "hdl64_structured", "os1_firing_real", "mixed", "hdl64_shuffled"):
    st = glob.glob(os.path.join(src, "trace_" + wl, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], f"{prefix}_{wl}_kernel_stats.csv")
    for name, out in ((f"bench_under_rocprof_{wl}.log", f"_{wl}_bench_under_rocprof.json"), (f"bench_{wl}.log", f"_{wl}_bench.json")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            json.dump(last_json_line(p), open(prefix + out, "w"), indent=1)

def pmc_summary(dir_glob, out_path, frames, what):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(src, dir_glob, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bevk::", "")
            if not k.startswith("k_") and "fillBuffer" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    if not agg:
        return
    out = {"source": f"rocprofv3 --pmc (one counter group per run) of `{what}` (BEV_LANES=1): counters summed over all "
                     f"dispatches of a kernel / {frames} frame passes",
           "frame_passes": frames,
           "corrections": "FETCH_SIZE is reported in KiB and counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md, "
                          "HBM): hbm_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE KiB is exact",
           "kernels": {}}
    total = 0.0
    for k in sorted(agg):
        m = {c: agg[k][c] / cnt[k][c] for c in agg[k]}   # per-dispatch means
        tot = {c: agg[k][c] for c in agg[k]}               # sums over the run
        e = {"dispatches": max(cnt[k].values())}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_read_bytes_per_launch"] = 2 * m["FETCH_SIZE"] * 1024
            e["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
            e["hbm_bytes_per_frame"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / frames
            e["hbm_read_bytes_per_frame"] = 2 * tot["FETCH_SIZE"] * 1024 / frames
            e["hbm_write_bytes_per_frame"] = tot["WRITE_SIZE"] * 1024 / frames
            e["FETCH_SIZE_KiB_raw"], e["WRITE_SIZE_KiB"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            if "fillBuffer" in k:
                e["note"] = "one-time clear of a winner table at context creation, not part of a step"
            else:
                total += e["hbm_bytes_per_frame"]
        if "TCC_EA0_RDREQ_sum" in m:
            e["TCC_EA0_RDREQ"], e["TCC_EA0_WRREQ"] = m["TCC_EA0_RDREQ_sum"], m.get("TCC_EA0_WRREQ_sum")
        if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_WAIT_ANY" in m and m.get("SQ_WAVE_CYCLES"):
            e["SQ_WAIT_ANY_over_WAVE_CYCLES"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        for a, b in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts"),
                     ("SQ_INSTS_LDS", "lds_insts")):
            if a in m:
                e[b] = m[a]
        out["kernels"][k] = e
    out["hbm_bytes_per_frame_all_kernels"] = total
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path, "total HBM bytes per frame:", round(total))


# bench.py --steps 1 --warmup 1 passes over its frames three times (warm-up, timed step, fenced step)
pmc_summary("pmc[0-9]*", prefix + "_pmc_traffic.json", pmc_frames, "bench.py --steps 1 --warmup 1 (the 1000-frame workload, sub-batch 500)")
pmc_summary("pmcgen[0-9]*", prefix + "_pmc_traffic_general.json", pmc_frames,
            "BEV_STREAM=0 bench.py --steps 1 --warmup 1 (the 1000-frame workload through the GENERAL path: order scan + gather walk)")
pmc_summary("pmcf[0-9]*", prefix + "_pmc_traffic_fused.json", pmc_frames,
            "bench.py --steps 1 --warmup 1 with FUSED launches (k_stage: the configuration of the timed region; no BEV_LANES=1)")
for wl in ("os1_firing", "hdl64_structured", "os1_firing_real", "hdl64_shuffled"):
    pmc_summary(f"pmc_{wl}[0-9]*", f"{prefix}_{wl}_pmc_traffic.json", pmc_frames, f"bench.py --steps 1 --warmup 1 --workload {wl}")
pmc_summary("pmc_oxford_concat[0-9]*", prefix + "_oxford_concat_pmc_traffic.json", pmc_frames // 10, "bench.py --steps 1 --warmup 1 --workload oxford_concat --frames 100")
# the bench lines of THIS run carry the traffic of THIS run's PMC passes (bench.py itself can only look at what was committed
# before it ran: the counters are collected after the bench line is printed)
PREFIX = {"k_walk": "k_walk<2,", "k_walk_general": "k_walk<0,", "k_walk_structured": "k_walk<3,", "k_walk_colmajor": "k_walk<4,",
          "k_walk_colmajor_gen": "k_walk<5,"}
for tag in ("", "os1_firing_", "hdl64_structured_", "os1_firing_real_", "oxford_concat_", "hdl64_shuffled_"):
    pmc_path = f"{prefix}_{tag}pmc_traffic.json"
    if not os.path.exists(pmc_path):
        continue
    pmc = json.load(open(pmc_path))
    for kind in ("bench", "bench_under_rocprof"):
        bp = f"{prefix}_{tag}{kind}.json"
        if not os.path.exists(bp):
            continue
        d = json.load(open(bp))
        r = d.get("roofline")
        if not r:
            continue
        pre = PREFIX.get(r["kernel"], r["kernel"])
        for kname, kv in pmc["kernels"].items():
            if kname.startswith(pre) and kv.get("hbm_bytes_per_frame", 0) > 1e5:
                r["traffic"] = kv["hbm_bytes_per_frame"] * r["frames_per_launch"]
                r["traffic_source"] = f"{os.path.basename(pmc_path)} (the PMC passes of the same profile_round.sh run; filled in by scripts/make_profiles.py): {pmc.get('source', '')}"
        tot = pmc.get("hbm_bytes_per_frame_all_kernels")
        if tot and "pipeline" in r:
            r["pipeline"]["hbm_traffic_per_frame_all_kernels"] = tot
            r["pipeline"]["real_traffic_gbps"] = tot * d["value"] / d["n_gpus"] / 1e9
        json.dump(d, open(bp, "w"), indent=1)
rp = os.path.join(src, "repeat.txt")
if os.path.exists(rp):
    shutil.copy(rp, prefix + "_repeat.txt")
print("files:", sorted(glob.glob(prefix + "_*")))
