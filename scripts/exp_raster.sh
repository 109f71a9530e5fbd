cd $GRAFT_REPO_ROOT
H=point-cloud-preprocessing-tools_amd/csrc/bev_internal.h
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; for lanes in 1 2; do BEV_LANES=$lanes timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1 lanes $lanes", round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/6)),2)) for k in d["kernels"]])
PY
done
}
run split4
sed -i 's/constexpr int kRasterSplit = 4; /constexpr int kRasterSplit = 8; /' $H; run split8
sed -i 's/constexpr int kRasterSplit = 8; /constexpr int kRasterSplit = 8; /; s/constexpr int kRasterThreads = 1024;/constexpr int kRasterThreads = 512;/' $H; run split8_512thr
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
