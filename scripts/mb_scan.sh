cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scanbw.hip -o /tmp/scanbw 2>/dev/null && timeout 120 /tmp/scanbw
