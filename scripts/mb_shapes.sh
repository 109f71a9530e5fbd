# what byte-moving kernels of different SHAPES reach on this box (scripts/microbench/copyshapes.hip), with the box's clocks
cd scripts/microbench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 copyshapes.hip -o /tmp/copyshapes && (rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -8; timeout -k 10 300 /tmp/copyshapes)
