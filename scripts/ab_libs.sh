# same-box A/B of several builds of the library through BEV_AMD_LIB: bash scripts/ab_libs.sh head mi355x a3 ...
# (csrc/libbev_<name>.so; "head" = a copy of an older build)
LIBS=${@:-head mi355x}
for rep in 1 2; do for lib in $LIBS; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_$lib.so timeout 300 python3 bench.py --no-build --steps 5 --warmup 2 --no-cpu $ABARGS 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$lib", round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/250,2)) for k in d["kernels"]])
PY
done; done
