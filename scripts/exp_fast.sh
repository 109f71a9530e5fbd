cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
cp $F /tmp/orig.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; BEV_FAST=1 BEV_LANES=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1", round(d["value"]), [(k["name"][2:13], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/4)),2)) for k in d["kernels"] if "strip" in k["name"] or "tail" in k["name"] or "prefix" in k["name"]])
PY
}
run base
sed -i 's#        return fl >= 0 ? bits\[fl >> 5\] : 0u;#        return 0u; /*EXP*/#' $F; run no_tail_word
cp /tmp/orig.hip $F
sed -i 's#            if (pos < 2 || pos >= 2 + kStripCols) { my_fail = true; return; }#            /*EXP*/#; s#            if (tid > 2 \&\& !(cprev != 0xffffffffu \&\& cprev < cs)) { my_fail = true; return; }#            /*EXP*/#' $F; run no_verify
cp /tmp/orig.hip $F
sed -i 's#        uint32_t cprev = __shfl_up(cs, 1);#        uint32_t cprev = 0; /*EXP*/#; s#        if (lane == 0 \&\& tid > 2 \&\& k.ci > 0 \&\& !k.wrap) cprev = point_slot(fpts, (uint32_t)k.ci - 1u, N, H);#        /*EXP*/#' $F; run no_cprev
cp /tmp/orig.hip $F
