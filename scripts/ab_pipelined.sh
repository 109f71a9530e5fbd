for rep in 1 2 3; do for lib in base mi355x; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_$lib.so timeout 300 python3 bench.py --no-build --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python3 - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); sb=d["config"]["sub_batch"]
print("$lib", round(d["value"]), "pipelined:", [(k["name"][2:], round(k["avg_launch_ms"]*1e3/sb,3)) for k in d["kernels_pipelined"] if k["avg_launch_ms"]*1e3/sb > 0.03])
PY
done; done
