# timing-only ablations of the in-place walk (results are WRONG: never used for anything but timing): what the walk costs without
# its ordered-cloud stores, code lists, candidate stores, status / candidate logic, BEV codes — cumulatively (a1..a5) and one at a
# time (b1..b3).  Build first (no GPU): for v in "a1:-DBEV_EXP_NOSTORE" "a2:-DBEV_EXP_NOSTORE -DBEV_ABL_NOLIST" ... ; do
#   make -C point-cloud-preprocessing-tools_amd exp EXPFLAGS="${v#*:}" EXPNAME=${v%%:*}; done   (profiles/r04_walk_ablation.txt)
for rep in 1 2; do for lib in mi355x a1 a2 a3 a4 a5 b1 b2 b3; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_$lib.so timeout 200 python3 bench.py --no-build --steps 8 --warmup 3 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python3 - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); sb=d["config"]["sub_batch"]
print("$lib", round(d["value"]), [(k["name"][2:], round(k["avg_launch_ms"]*1e3/sb,3)) for k in d["kernels"] if k["avg_launch_ms"]*1e3/sb > 0.03])
PY
done; done
