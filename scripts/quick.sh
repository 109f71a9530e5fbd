# usage (on the GPU box): bash scripts/quick.sh <tag> [pytest args...]  -> parity subset, bench line, kernel summary
# stops at the first failing step (no GPU step is started after a failed one)
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-q}; shift
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py -m gpu -x -q "$@" > gpurun_out/${TAG}_parity.log 2>&1 || { tail -20 gpurun_out/${TAG}_parity.log; exit 1; }
tail -2 gpurun_out/${TAG}_parity.log
timeout -k 10 600 python3 bench.py --no-build --steps 10 --warmup 3 --no-cpu > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${TAG}_bench.log") if l.startswith("{")][-1])
F=d["config"]["frames_per_gpu"]
print(round(d["value"]), "frames/s; dominant-kernel frac", round(d["roofline"]["frac"],3), "pipeline frac", round(d["roofline"]["pipeline"]["frac"],3))
print(" one-lane  us/frame:", [(k["name"][2:], round(k["avg_launch_ms"]*k["launches"]/d["steps"]/F*1e3,3)) for k in d["kernels"]])
print(" pipelined us/frame:", [(k["name"][2:], round(k["total_ms"]/d["steps"]/F*1e3,3)) for k in d["kernels_pipelined"]])
PY
