# usage (on the GPU box): bash scripts/pmc_insts.sh <tag> ["ENV=V ..."]  -> per-kernel instruction counts per frame
# (one rocprofv3 PMC pass, no trace domains; the libraries must already be built: nothing is spawned under the profiler)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-insts}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for kv in $2; do export "$kv"; done
BEV_LANES=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/pmc -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'bevk' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
frames=3000.0  # 1 warm-up + 1 timed step + 1 fenced step of 1000 frames
for k,v in agg.items():
    w=v.get('SQ_WAVES',1)
    print(f"{k[:34]:34s} per frame: valu {v['SQ_INSTS_VALU']/frames/1e3:7.1f}k salu {v['SQ_INSTS_SALU']/frames/1e3:7.1f}k lds {v['SQ_INSTS_LDS']/frames/1e3:6.1f}k vmem_rd {v['SQ_INSTS_VMEM_RD']/frames/1e3:5.1f}k wr {v['SQ_INSTS_VMEM_WR']/frames/1e3:5.1f}k waves {w/frames:6.1f}  valu/wave {v['SQ_INSTS_VALU']/w:7.0f}")
PY
