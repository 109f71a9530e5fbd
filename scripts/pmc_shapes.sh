# usage (GPU box): bash scripts/pmc_shapes.sh <tag>   -> TLB and memory-side stall counters of the copy shapes (copyshapes.hip):
# is a looping, many-streams kernel (the walk's shape) slower than a no-loop copy because of address translation or because
# of the memory side?  One counter group per pass, no trace domains.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-pmcshapes}; mkdir -p $OUT
cd $R/scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 copyshapes.hip -o /tmp/copyshapes || exit 1
cd /tmp
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- /tmp/copyshapes > $OUT/p$i.log 2>&1 || { tail -3 $OUT/p$i.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in sorted(agg):
    print(k)
    for c,v in sorted(agg[k].items()): print(f"    {c:44s} per launch {v/cnt[k][c]:.4g}")
PY
