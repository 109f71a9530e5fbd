# usage (GPU box): bash scripts/fr_traffic.sh  -> HBM bytes read by the general firing-order walk per OS1_64 frame, with and without
# no-return records / staggered beams / backward rotation (bench.py developer knobs BEV_FR_NORET, BEV_FR_STAGGER, BEV_FR_PHASE, BEV_FR_DIR)
export TMPDIR=/tmp
for cfg in "default" "BEV_FR_NORET=0" "BEV_FR_STAGGER=0" "BEV_FR_NORET=0 BEV_FR_STAGGER=0 BEV_FR_PHASE=5 BEV_FR_DIR=1" "BEV_FR_NORET=0 BEV_FR_STAGGER=0 BEV_FR_PHASE=5 BEV_FR_DIR=-1"; do
  echo "== $cfg"
  ( for kv in $cfg; do [ "$kv" != default ] && export "$kv"; done
    rm -rf gpurun_out/frp; mkdir -p gpurun_out/frp
    BEV_LANES=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/frp/pmc -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile --workload os1_firing_real > gpurun_out/frp/log 2>&1 || tail -3 gpurun_out/frp/log
    python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob('gpurun_out/frp/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_walk' in k: agg[k]+=float(r['Counter_Value'])
for k,v in agg.items(): print("  ", k[:40], "read MB/frame", round(2*v*1024/3000/1e6,3))
PY
  )
done
