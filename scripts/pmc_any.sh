# usage (on the GPU box): bash scripts/pmc_any.sh <tag> "<counter> <counter> ..." ["ENV=V ..."]
#   one rocprofv3 PMC pass (no trace domains) of the 1000-frame bench with one lane; prints the counters per kernel,
#   per frame.  The libraries must already be built (nothing is spawned under the profiler).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-pmc}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for kv in $3; do export "$kv"; done
BEV_LANES=1 timeout -k 10 300 rocprofv3 --pmc $2 --output-format csv -d $OUT/pmc -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'bevk' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(f"{k[:34]:34s}", " ".join(f"{c[3:]}={x/3000.0/1e3:.1f}k" for c,x in sorted(v.items())))
PY
