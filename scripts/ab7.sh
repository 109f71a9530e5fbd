# phase clocks of k_cell_sums (developer build, make clk)
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_mi355x_clk.so BEV_LANES=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-profile 2>&1 | grep "^cell_sums" | head -6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
for lanes in 1 2; do
BEV_LANES=$lanes timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("lanes $lanes", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
