# same-box A/B of two builds of the library over several workloads (round 6: profiles/r06_ab_vs_r05.txt):
#   bash scripts/build_rev_lib.sh <git-rev> old        (no GPU needed)
#   bash scripts/ab_workloads.sh old mi355x hdl64_sweep hdl64_structured os1_firing os1_firing_real mixed      (GPU box)
# three interleaved passes per workload, bench.py --steps 20 --warmup 5, the library selected with BEV_AMD_LIB
R=$GRAFT_REPO_ROOT; cd $R
A=$1; B=$2; shift 2
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
for wl in "$@"; do for rep in 1 2 3; do for lib in $A $B; do
BEV_AMD_LIB=$R/point-cloud-preprocessing-tools_amd/csrc/libbev_$lib.so timeout 300 python3 bench.py --no-build --steps 20 --warmup 5 --no-cpu --no-profile --workload $wl 2>/dev/null | tail -1 > /tmp/b.json || exit 1
python3 - $lib $wl <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read())
print(sys.argv[2], sys.argv[1], round(d["value"]), "frames/s; fenced median", round(d["ms_per_step_median_fenced"], 3), "ms")
PY
done; done; done
