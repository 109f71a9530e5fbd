cd $GRAFT_REPO_ROOT
F=point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip
cp $F /tmp/orig.hip
run() { make -C point-cloud-preprocessing-tools_amd 2>&1 | grep -E "error" ; BEV_LANES=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); print("$1", round(d["value"]), [(k["name"][2:8], round(k["avg_launch_ms"]*1e3/ (1000/ (k["launches"]/5)),2)) for k in d["kernels"]])
PY
}
run base
python - <<'PY'
p='point-cloud-preprocessing-tools_amd/csrc/bev_kernels.hip'; s=open(p).read()
a=s.index('                    const int st = s0 + k;\n                    const uint2 aux = b.cand_aux')
b=s.index('                    b.codes[idx] = aux.y;\n')+len('                    b.codes[idx] = aux.y;\n')
s=s[:a]+'                    b.ncand[0] = 0; /* keep the test alive, one address */\n'+s[b:]
open(p,'w').write(s)
PY
run no_body
cp /tmp/orig.hip $F
