timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in "hdl64_sweep --frames 1000" "os1_firing --frames 1000"; do timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print(d["config"]["workload"][:40], "|", round(d["value"]), "fps", [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (d["config"]["frames_per_gpu"]/ (k["launches"]/5)),2)) for k in d["kernels"]])
PY
done
