cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
cp $F /tmp/orig.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; BEV_LANES=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1", [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/4)),2)) for k in d["kernels"] if "cell" in k["name"]])
PY
}
run full
sed -i 's#    /\* per-cell totals, hist -> wave offsets inside the cell.s run \*/#    if (blockIdx.x >= 0) return; /*EXP after pass1*/#' $F; run after_pass1
cp /tmp/orig.hip $F
sed -i 's#    /\* pass 2: stable placement.  Segments are walked IN ORDER; the (cell, z) pairs of the next#    if (blockIdx.x >= 0) return; /*EXP after scan*/ /*#' $F; run after_scan
cp /tmp/orig.hip $F
sed -i 's#    /\* pass 3: in-order float accumulation; thread owns cells tid + 512\*j \*/#    if (blockIdx.x >= 0) return; /*EXP after pass2*/#' $F; run after_pass2
cp /tmp/orig.hip $F
