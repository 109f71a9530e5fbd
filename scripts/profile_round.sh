# usage (on the GPU box): bash scripts/profile_round.sh <tag> [pmc] [pmcothers] [others] [repeat]   -> gpurun_out/<tag>/...
#   default : plain bench line (with CPU baseline) + rocprofv3 kernel trace/stats of the same command with one lane
#   pmc     : + the PMC passes (one counter group per run) on the SAME 1000-frame workload
#   pmcothers: + FETCH_SIZE / WRITE_SIZE passes of the general path (BEV_STREAM=0) and of os1_firing / hdl64_structured
#   others  : + bench line and kernel stats of BASELINE configs 3 (os1_firing) and 5 (oxford_concat), of hdl64_structured, os1_firing_real, mixed and hdl64_shuffled (the general path)
#   repeat  : + the graded workload and hdl64_structured five times each, unprofiled, on the same box -> repeat.txt
# The libraries are built ONCE up front; every profiled command is `rocprofv3 ... -- python3 bench.py --no-build`, so
# nothing is spawned from a process the profiler has already attached to the GPU.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r02}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
# 1. plain bench first (with the CPU baseline): the GPU slows down by 5-10 % once the profiled runs have warmed it up
timeout 900 python3 bench.py --no-build --steps 20 --warmup 5 > $OUT/bench.log 2>$OUT/bench.err || exit 1
tail -c 600 $OUT/bench.log
# 2. kernel trace + stats of the bench command line with serial launches (BEV_LANES=1): every kernel a launch of its own, back to back, like in
#    bench.py's roofline pass, so AverageNs is comparable with roofline.avg_launch_ms
BEV_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-build --steps 5 --warmup 2 --no-cpu > $OUT/bench_under_rocprof.log 2>$OUT/bench_under_rocprof.err || exit 1
for w in "$@"; do
if [ "$w" = pmc ]; then
# 3. PMC passes, one counter group per run (no trace domains mixed in); 1 warm-up + 1 step = 2000 frame passes
ARGS="bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  BEV_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || exit 1
done
# 3a. the same workload as the timed region runs it — fused launches (k_stage), no BEV_LANES=1: the bytes of the WHOLE path in
#     the configuration the headline comes from (one kernel name: the per-stage split is the serial passes' above)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcf$i -- python3 $ARGS > $OUT/pmcf$i.log 2>&1 || exit 1
done
fi
if [ "$w" = pmcothers ]; then
# 3b. HBM traffic (FETCH_SIZE, WRITE_SIZE: two passes each) of the general path on the headline workload and of the other layouts
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  BEV_STREAM=0 BEV_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcgen$i -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile > $OUT/pmcgen$i.log 2>&1 || exit 1
  for wl in os1_firing hdl64_structured os1_firing_real oxford_concat hdl64_shuffled; do
    F=1000; if [ $wl = oxford_concat ]; then F=100; fi
    BEV_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${wl}$i -- python3 bench.py --no-build --steps 1 --warmup 1 --no-cpu --no-profile --workload $wl --frames $F > $OUT/pmc_${wl}$i.log 2>&1 || exit 1
  done
done
fi
if [ "$w" = repeat ]; then
# 3c. the graded workload and the structured layout five times each, unprofiled, interleaved: median and spread on ONE box
for rep in 1 2 3 4 5; do for wl in hdl64_sweep hdl64_structured; do
  timeout 300 python3 bench.py --no-build --steps 20 --warmup 5 --no-cpu --workload $wl 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['value']), 'frames/s; fenced median', round(1e3*d['config']['frames_per_gpu']/d['ms_per_step_median_fenced']), '; dominant kernel frac', round(d['roofline']['frac'],3), 'pipeline frac', round(d['roofline']['pipeline']['frac'],3))" >> $OUT/repeat.txt || exit 1
done; done
fi
if [ "$w" = others ]; then
for wl in os1_firing oxford_concat hdl64_structured os1_firing_real mixed hdl64_shuffled; do
  F=1000; if [ $wl = oxford_concat ]; then F=100; fi
  timeout 900 python3 bench.py --no-build --steps 20 --warmup 5 --workload $wl --frames $F --cpu-sample 50 > $OUT/bench_$wl.log 2>$OUT/bench_$wl.err || exit 1
  BEV_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- python3 bench.py --no-build --steps 3 --warmup 1 --no-cpu --workload $wl --frames $F > $OUT/bench_under_rocprof_$wl.log 2>&1 || exit 1
done
fi
done
echo profile_round done
