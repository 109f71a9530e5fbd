# phase clocks of k_cell_sums (developer build, make clk)
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/libbev_mi355x_clk.so BEV_LANES=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-profile 2>&1 | grep "^cell_sums" | head -6
