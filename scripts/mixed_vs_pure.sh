# usage (GPU box): bash scripts/mixed_vs_pure.sh  -> the mixed workload against its three pure ones on ONE box, twice over
v() { python3 bench.py --no-build --steps 20 --warmup 5 --no-cpu "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config'].get('routes_last_sub_batch'))"; }
for rep in 1 2; do
echo "hdl64_sweep $(v --workload hdl64_sweep)"
echo "hdl64_structured $(v --workload hdl64_structured)"
echo "os1_firing_real@HDL_64E $(v --workload os1_firing_real --sensor HDL_64E)"
echo "mixed $(v --workload mixed)"
done
