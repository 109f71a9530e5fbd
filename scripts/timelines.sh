# usage (GPU box): bash scripts/timelines.sh  -> gpurun_out/tl/*.txt: the pipeline on a time line (one call of one sub-batch, and ten calls back to back: fused launches on two streams; HDL_64E and OS1_64) and k_cell_sums by workgroup
# (developer builds: make -C point-cloud-preprocessing-tools_amd tl cstl cstl0)
set -e
L=$PWD/point-cloud-preprocessing-tools_amd/csrc
mkdir -p gpurun_out/tl
BEV_AMD_LIB=$L/libbev_tl_all.so TL_CALLS=1 timeout -k 10 300 python3 scripts/pipeline_timeline.py 1000 HDL_64E 50 1000 > gpurun_out/tl/hdl_serial.txt 2>&1
BEV_AMD_LIB=$L/libbev_tl_all.so TL_CALLS=10 timeout -k 10 300 python3 scripts/pipeline_timeline.py 1000 HDL_64E 25 500 > gpurun_out/tl/hdl_pipelined.txt 2>&1
BEV_AMD_LIB=$L/libbev_tl_all.so TL_CALLS=10 timeout -k 10 300 python3 scripts/pipeline_timeline.py 1000 OS1_64 25 500 > gpurun_out/tl/os1_pipelined.txt 2>&1
BEV_AMD_LIB=$L/libbev_tl_all.so TL_CALLS=1 timeout -k 10 300 python3 scripts/pipeline_timeline.py 1000 OS1_64 50 1000 > gpurun_out/tl/os1_serial.txt 2>&1
BEV_AMD_LIB=$L/libbev_cstl.so timeout -k 10 300 python3 scripts/cell_sums_timeline.py OS1_64 2>&1 | grep -v "^cell_sums barrier0\|^cell_sums all-parts\|^walk\|^raster\|^probe" > gpurun_out/tl/cs_os1.txt
BEV_AMD_LIB=$L/libbev_cstl0.so timeout -k 10 300 python3 scripts/cell_sums_timeline.py OS1_64 2>&1 | grep -v "^cell_sums barrier0\|^cell_sums all-parts\|^walk\|^raster\|^probe" > gpurun_out/tl/cs_os1_unrotated.txt
BEV_AMD_LIB=$L/libbev_cstl.so timeout -k 10 300 python3 scripts/cell_sums_timeline.py HDL_64E 2>&1 | grep -v "^cell_sums barrier0\|^cell_sums all-parts\|^walk\|^raster\|^probe" > gpurun_out/tl/cs_hdl.txt
head -8 gpurun_out/tl/os1_pipelined.txt gpurun_out/tl/os1_serial.txt; grep -c . gpurun_out/tl/*.txt
