export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BEV_LANES=${LANES:-2} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_l2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-profile --sub-batch 256 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/trace_l2/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'bevk' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]   # steady state (last step)
t0=int(rows[0]['Start_Timestamp'])
iv=[(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0,r) for r in rows]
# union length and overlap
ev=sorted([(a,1) for a,b,_ in iv]+[(b,-1) for a,b,_ in iv])
cur=0; last=0; busy1=0; busy2=0
for t,d in ev:
    if cur>=1: busy1+=t-last
    if cur>=2: busy2+=t-last
    cur+=d; last=t
span=max(b for a,b,_ in iv)
print("span %.0f us, >=1 kernel running %.0f us, >=2 running %.0f us, sum of durations %.0f us"%(span/1e3,busy1/1e3,busy2/1e3,sum(b-a for a,b,_ in iv)/1e3))
for a,b,r in iv[:24]:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('bevk::','')[:18]
    print(r['Queue_Id'], n.ljust(20), "%.0f -> %.0f  (%.0f us)"%(a/1e3,b/1e3,(b-a)/1e3))
PY
