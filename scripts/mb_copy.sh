# mixed read/write bandwidth ceilings of the box (scripts/microbench/copybw.hip)
cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 copybw.hip -o /tmp/copybw && timeout -k 10 120 /tmp/copybw
