# usage: bash scripts/knob.sh "ENV=VAL ..." ... ; one bench line per setting (same box, interleaved twice)
#        BENCH_ARGS=--sub-batch,128 inside a setting adds bench.py arguments (commas for blanks)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
for setting in "$@"; do
  env $setting timeout -k 10 300 python3 bench.py --no-build --steps 8 --warmup 3 --no-cpu $(for kv in $setting; do case $kv in BENCH_ARGS=*) echo ${kv#BENCH_ARGS=} | tr , " ";; esac; done) 2>/dev/null | tail -1 > /tmp/knob.json || exit 1
  python3 - "$setting" <<'PY'
import json,sys
d=json.loads(open("/tmp/knob.json").read()); F=d["config"]["frames_per_gpu"]
print(f"{sys.argv[1]:28s}", round(d["value"]), "pipe us/frame:", [(k["name"][2:8], round(k["total_ms"]/d["steps"]/F*1e3,2)) for k in d["kernels_pipelined"]], "one-lane:", [round(k["avg_launch_ms"]*k["launches"]/d["steps"]/F*1e3,2) for k in d["kernels"]])
PY
done; done
