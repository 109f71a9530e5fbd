# does line alignment of the rows matter for the walk?  H = 2083 (rows start 0/96/64/32 B into a line) vs 2084 (aligned)
for H in ${HS:-2083 2084 2080}; do
sed "s/    p = bev_amd.params_for_sensor(args.sensor)/    p = bev_amd.params_for_sensor(args.sensor); p.horizon_scan = $H/" bench.py > /tmp/bench_h.py
cp /tmp/bench_h.py ./bench_h_tmp.py
BEV_LANES=1 timeout 300 python bench_h_tmp.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("H $H", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
rm -f bench_h_tmp.py
