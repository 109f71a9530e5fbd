# usage: bash scripts/pmc.sh <outdir-name> ; collects HBM/L2/SQ counters in separate passes (no trace domains mixed in)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu --no-profile --frames 256 --sub-batch 256"
export BEV_LANES=1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bevk::","")
        if not k.startswith("k_") and "fillBuffer" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in agg:
    print(k)
    for c,v in sorted(agg[k].items()):
        n=cnt[k][c]; print("   %-40s per-launch %.4g  (launches %d)"%(c, v/n, n))
PY
