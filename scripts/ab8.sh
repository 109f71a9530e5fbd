timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
for cfg in "0 2" "1 2" "1 3" "0 3" "0 2" "1 2"; do set -- $cfg
BEV_STAGED=$1 BEV_LANES=$2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-profile 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("staged $1 lanes $2", round(d["value"]))
PY
done
BEV_STAGED=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --workload os1_firing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('os1', round(d['value']))"
