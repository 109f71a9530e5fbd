# throughput and per-kernel one-lane times against the sub-batch size: bash scripts/sweep_sb.sh [lib name] (csrc/libbev_<name>.so)
LIB=${1:-mi355x}
for sb in 32 64 96 128 192 256 334 500; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_$LIB.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --sub-batch $sb 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); F=d["config"]["frames_per_gpu"]
print("$LIB sb", d["config"]["sub_batch"], round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3*k["launches"]/d["steps"]/F,2)) for k in d["kernels"]])
PY
done
