# same-box A/B of several builds of the library through BEV_AMD_LIB: bash scripts/ab_libs.sh head mi355x a3 ...
# (csrc/libbev_<name>.so; scripts/build_rev_lib.sh builds one from a git revision; ABARGS="--workload os1_firing" etc.)
LIBS=${@:-head mi355x}
for rep in 1 2 3; do for lib in $LIBS; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_$lib.so timeout 300 python3 bench.py --no-build --steps 10 --warmup 3 --no-cpu $ABARGS 2>/dev/null | tail -1 > /tmp/b.json; python3 - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); sb=d["config"]["sub_batch"]
print("$lib", round(d["value"]), [(k["name"][2:], round(k["avg_launch_ms"]*1e3/sb,3)) for k in d["kernels"] if k["avg_launch_ms"]*1e3/sb > 0.03])
PY
done; done
