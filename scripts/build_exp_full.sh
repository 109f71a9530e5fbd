# usage: bash scripts/build_exp_full.sh <name> <flags...>  -> csrc/libbev_<name>.so with BOTH translation units compiled
# with the flags (make exp recompiles the kernels only)
cd "$(dirname "$0")/../point-cloud-preprocessing-tools_amd" || exit 1
name=$1; shift
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FL "$@" -c csrc/bev_kernels.hip -o csrc/bev_kernels_$name.o &
/opt/rocm/bin/hipcc $FL "$@" -c csrc/bev_capi.hip -o csrc/bev_capi_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o csrc/libbev_$name.so csrc/bev_kernels_$name.o csrc/bev_capi_$name.o -L/opt/rocm/lib -lroctx64 -Wl,-rpath,/opt/rocm/lib && echo built libbev_$name.so
