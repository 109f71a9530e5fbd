for lanes in 1 2; do for sb in 200 227 256 341 455; do
BEV_LANES=$lanes timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --sub-batch $sb 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("lanes $lanes sb", d["config"]["sub_batch"], round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/3)),2)) for k in d["kernels"]])
PY
done; done
