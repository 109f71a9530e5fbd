cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; BEV_LANES=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/5)),2)) for k in d["kernels"] if "order" in k["name"]])
PY
}
for b in 16 8 32 64; do sed -i "s/constexpr int kScanBlocksPerFrame = [0-9]*;/constexpr int kScanBlocksPerFrame = $b;/" $F; run blocks_$b; done
sed -i "s/constexpr int kScanBlocksPerFrame = [0-9]*;/constexpr int kScanBlocksPerFrame = 16;/; s/constexpr int kScanPerThread = 8; /constexpr int kScanPerThread = 4; /" $F; run u4_blocks16
