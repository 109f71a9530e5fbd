BEV_LANES=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-profile 2>&1 | grep "^cs \|^p2" | head -24
