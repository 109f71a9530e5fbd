export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lanes in 2 3; do
BEV_LANES=$lanes timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-profile --sub-batch 64 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noprofile lanes $lanes', round(d['value']))"
done
BEV_LANES=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_l2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-profile --sub-batch 64 --frames 512 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_l2/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'bevk' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-40:]:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('bevk::','')[:22]
    print(r['Queue_Id'], r.get('Stream_Id'), n.ljust(24), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3)
PY
