cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
cp $F /tmp/orig.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; BEV_LANES=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1", [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/4)),2)) for k in d["kernels"] if "order" in k["name"] or "strip" in k["name"]])
PY
}
run base
sed -i 's#            atomicMax(&fw\[row \* H + col\], i + 1u);#            fw[row * H + col] = i + 1u; /*EXP plain store*/#' $F; run plain_store
cp /tmp/orig.hip $F
sed -i 's#            atomicMax(&fw\[row \* H + col\], i + 1u);#            if (i == 0xfffffff0u) fw[row * H + col] = i + 1u; /*EXP none*/#' $F; run no_write
cp /tmp/orig.hip $F
