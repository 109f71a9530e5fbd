# usage (GPU box): bash scripts/pmc_sq.sh <tag> <workload> ["ENV=V ..."]  -> SQ cycle / instruction counters per kernel per frame
# (three PMC passes, one lane, the 1000-frame bench; no trace domains)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=$1; WL=$2; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
for kv in $3; do export "$kv"; done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  BEV_LANES=1 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile --workload $WL > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'bevk' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if v.get('SQ_WAVE_CYCLES',0)/3000 < 1e4: continue
    print(k[:60])
    for c,x in sorted(v.items()): print(f"    {c:24s} {x/3000.0/1e3:10.1f} k per frame")
PY
