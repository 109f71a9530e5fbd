# usage (on the GPU box): bash scripts/profile_round.sh <tag>   -> gpurun_out/<tag>/...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
# 1. plain bench first (with the CPU baseline): the GPU slows down by 5-10 % once the profiled runs have warmed it up
timeout 900 python3 bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1
# 2. kernel trace + stats of the bench command line with ONE lane (BEV_LANES=1): every launch runs back to back, like in
#    bench.py's roofline pass, so AverageNs is comparable with roofline.avg_launch_ms
BEV_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
# 3. PMC passes, one counter group per run (no trace domains mixed in)
ARGS="bench.py --steps 1 --warmup 1 --no-cpu --no-profile --frames 256 --sub-batch 256"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  BEV_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1
done
tail -1 $OUT/bench.log
