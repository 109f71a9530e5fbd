cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 readbw.hip -o /tmp/readbw && timeout 120 /tmp/readbw
