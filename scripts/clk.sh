# phase clocks inside the kernels (developer build: make -C point-cloud-preprocessing-tools_amd clk); usage: clk.sh [regex]
R=$GRAFT_REPO_ROOT; cd $R
BEV_AMD_LIB=$R/point-cloud-preprocessing-tools_amd/csrc/libbev_mi355x_clk.so BEV_LANES=${LANES:-1} timeout -k 10 300 python3 bench.py --no-build --steps 2 --warmup 0 --no-cpu --no-profile 2>&1 | grep -E "^(${1:-cell_sort|cell_sum|raster|walk})" | sort | uniq -c | sort -rn | head -${2:-12}
