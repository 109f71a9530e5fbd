# row-block tile shapes against a no-loop copy (scripts/microbench/tileshapes.hip)
cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tileshapes.hip -o /tmp/tileshapes && timeout -k 10 300 /tmp/tileshapes
