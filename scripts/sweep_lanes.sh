# throughput against the number of workspace sets (BEV_LANES) and the sub-batch size, staged pipeline
for rep in 1 2; do for lanes in 2 3 4; do for sb in 250 256; do
BEV_LANES=$lanes timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-profile --sub-batch $sb 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes sb $sb', round(d['value']))"
done; done; done
