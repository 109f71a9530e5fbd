cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scatterw.hip -o /tmp/scatterw && timeout 120 /tmp/scatterw
