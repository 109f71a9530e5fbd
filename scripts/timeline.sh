# kernel timeline (start/end per dispatch) of the default run (fused launches on two streams), to see which launches really overlap
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --no-build --steps 3 --warmup 1 --no-cpu --no-profile > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/timeline/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id","?"), r.get("Stream_Id","?")) for r in rows]
ev.sort()
t0 = ev[0][0]
# keep the last third (steady state)
ev = [e for e in ev if "bevk" in e[2] or "fill" in e[2].lower() or "memset" in e[2].lower()]
tail = ev[len(ev)*2//3:]
out = open("gpurun_out/timeline/timeline.txt", "w")
for s, e, n, q, st in tail:
    out.write(f"{(s-t0)/1e3:10.1f} {(e-t0)/1e3:10.1f} {(e-s)/1e3:8.1f} us  q{q} s{st} {n}\n")
# overlap summary: for each kernel name, mean duration
d = collections.defaultdict(list)
for s, e, n, q, st in tail: d[n].append((e-s)/1e3)
for n, v in d.items(): out.write(f"# {n}: n={len(v)} mean {sum(v)/len(v):.1f} us\n")
span = (tail[-1][1]-tail[0][0])/1e3
busy = sum(e-s for s,e,*_ in tail)/1e3
out.write(f"# span {span:.1f} us, sum of durations {busy:.1f} us, concurrency {busy/span:.2f}\n")
PY
tail -12 $OUT/timeline.txt; tail -1 $OUT/bench.log | cut -c1-200
