timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py tests/test_gpu_scale.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3
for cfg in "1 1" "1 2" "1 2"; do set -- $cfg
BEV_STAGED=$1 BEV_LANES=$2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("staged $1 lanes $2", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
STAGED=1 bash scripts/timeline.sh > /dev/null 2>&1; tail -8 gpurun_out/timeline/timeline.txt
