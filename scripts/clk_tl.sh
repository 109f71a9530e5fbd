# when and where the workgroups of ONE in-place walk launch ran (developer build: make clk): bash scripts/clk_tl.sh [frames]
R=$GRAFT_REPO_ROOT; cd $R
F=${1:-128}
BEV_AMD_LIB=$R/point-cloud-preprocessing-tools_amd/csrc/libbev_mi355x_clk.so BEV_LANES=1 timeout -k 10 300 python3 bench.py --no-build --steps 1 --warmup 0 --no-cpu --no-profile --frames $F --sub-batch $F 2>&1 | grep -E "^walk_tl" > gpurun_out/walk_tl_$F.txt
python3 - gpurun_out/walk_tl_$F.txt <<'PY'
import sys,collections
rows=[l.split() for l in open(sys.argv[1])]
rows=[(int(r[1]),int(r[2]),int(r[3]),int(r[4]),int(r[5])) for r in rows]
# the bench launches the walk more than once (probe pass etc.): keep the last launch = the latest block 0
t_first=max(r[3] for r in rows if r[0]==0)
rows=[r for r in rows if r[3]>=t_first-5]
t0=min(r[3] for r in rows); t1=max(r[4] for r in rows)
print(len(rows),"workgroups sampled; launch spans",(t1-t0)/100.0,"us")
life=[(r[4]-r[3])/100.0 for r in rows]; starts=sorted((r[3]-t0)/100.0 for r in rows)
print("lifetime us: min %.1f median %.1f max %.1f"%(min(life),sorted(life)[len(life)//2],max(life)))
print("start offsets us (deciles):",[round(starts[int(i*(len(starts)-1)/10)],1) for i in range(11)])
percu=collections.Counter()
for b,hw,xcc,a,e in rows:
    percu[(xcc&0xf,(hw>>13)&7,(hw>>12)&1,(hw>>8)&0xf)]+=1
print("distinct (xcc,se,sh,cu):",len(percu),"sampled WGs per CU: min",min(percu.values()),"max",max(percu.values()))
PY
