# usage (on the GPU box): bash scripts/walk_latency.sh ["ENV=V ..."] -> duration of ONE column-walk launch at 1, 2, 4 and 8 workgroups per CU
# (frames per launch 32 .. 256): is a row step bounded by its own latency chain or by the CU's throughput?
R=$GRAFT_REPO_ROOT; cd $R
for kv in $1; do export "$kv"; done
for F in 32 64 96 128 256; do
  BEV_LANES=1 timeout -k 10 300 python3 bench.py --no-build --steps 10 --warmup 3 --no-cpu --frames $F --sub-batch $F 2>/dev/null | tail -1 > /tmp/wl.json || exit 1
  python3 - $F <<'PY'
import json,sys
d=json.loads(open("/tmp/wl.json").read()); F=int(sys.argv[1])
for k in d["kernels"]:
    if k["name"] in ("k_walk", "k_walk_general") or "cell_sums" in k["name"]:
        print(F, "frames:", k["name"], "launch", round(k["avg_launch_ms"]*1e3,1), "us ->", round(k["avg_launch_ms"]*1e3/F,2), "us/frame;", "launches", k["launches"])
PY
done
