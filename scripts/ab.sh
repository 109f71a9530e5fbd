timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for lanes in 1 2; do
BEV_LANES=$lanes timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --sub-batch ${SB:-256} 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("lanes $lanes sb", d["config"]["sub_batch"], round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
