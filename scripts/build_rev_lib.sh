# usage: bash scripts/build_rev_lib.sh <git-rev> <name>  -> point-cloud-preprocessing-tools_amd/csrc/libbev_<name>.so built from that
# revision's csrc/ and include/ (EXTRA="-DBEV_CS_CLOCK" etc. adds compiler flags) (for same-box A/B runs through BEV_AMD_LIB, scripts/ab_libs.sh); no GPU needed
set -e
REV=$1; NAME=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pkg/csrc $T/include
# (every tracked source of that revision's csrc/: one translation unit over per-kernel headers since round 6)
for f in $(git -C $R ls-tree --name-only $REV point-cloud-preprocessing-tools_amd/csrc/ | grep -E '\.(hip|h)$'); do git -C $R show $REV:$f > $T/pkg/csrc/$(basename $f); done
git -C $R show $REV:include/bev_mi355x.h > $T/include/bev_mi355x.h
FLAGS="$EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
(cd $T/pkg && /opt/rocm/bin/hipcc $FLAGS -c csrc/bev_kernels.hip -o k.o && /opt/rocm/bin/hipcc $FLAGS -c csrc/bev_capi.hip -o c.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/point-cloud-preprocessing-tools_amd/csrc/libbev_$NAME.so k.o c.o -L/opt/rocm/lib -lroctx64 -Wl,-rpath,/opt/rocm/lib)
rm -rf $T
echo built csrc/libbev_$NAME.so from $REV
