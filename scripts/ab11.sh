for rep in 1 2 3; do for sg in 0 1; do
BEV_STAGED=$sg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('staged $sg', round(d['value']))"
done; done
