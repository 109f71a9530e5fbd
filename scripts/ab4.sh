BEV_FAST=1 timeout 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for fast in 1 0; do for lanes in 1 2; do
BEV_FAST=$fast BEV_LANES=$lanes timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --sub-batch 256 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("fast $fast lanes $lanes", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done; done
