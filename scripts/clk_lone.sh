# phase clocks of the in-place walk for one workgroup per CU (32 frames) and for the full machine (1000 frames): where a
# row step's LATENCY goes, and what contention adds (developer build: make -C point-cloud-preprocessing-tools_amd clk)
R=$GRAFT_REPO_ROOT; cd $R
for F in 32 64 128 1000; do
  SB=$F; [ $F -gt 500 ] && SB=500
  echo "== $F frames per launch"
  BEV_AMD_LIB=$R/point-cloud-preprocessing-tools_amd/csrc/libbev_mi355x_clk.so BEV_LANES=1 timeout -k 10 300 python3 bench.py --no-build --steps 1 --warmup 0 --no-cpu --no-profile --frames $F --sub-batch $SB 2>&1 | grep -E "^walk" | head -4
done
