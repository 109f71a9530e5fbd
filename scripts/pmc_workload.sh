# usage (GPU box): bash scripts/pmc_workload.sh <tag> <workload> <frames> "<counter> ..."  -> per kernel per frame (one PMC pass, one lane)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=$1; WL=$2; F=$3; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
BEV_LANES=1 timeout -k 10 300 rocprofv3 --pmc $4 --output-format csv -d $OUT/pmc -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile --workload $WL --frames $F > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - $OUT $F <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'bevk' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
passes=3.0*float(sys.argv[2])
for k,v in agg.items():
    print(f"{k[:34]:34s}", " ".join(f"{c}={x/passes/1e3:.1f}k" for c,x in sorted(v.items())))
PY
