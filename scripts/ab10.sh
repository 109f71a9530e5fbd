BEV_LANES=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print(round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
