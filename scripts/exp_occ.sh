cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; for lanes in 1 2 3; do BEV_LANES=$lanes timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1 lanes $lanes", round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
}
run vgpr84
sed -i 's/__global__ __launch_bounds__(kStripThreads) void k_strip_ground/__global__ __launch_bounds__(kStripThreads, 6) void k_strip_ground/' $F; run bounds6
sed -i 's/__global__ __launch_bounds__(kStripThreads, 6) void k_strip_ground/__global__ __launch_bounds__(kStripThreads, 7) void k_strip_ground/' $F; run bounds7
