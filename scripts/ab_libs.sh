# same-box A/B of two builds of the library: csrc/libbev_head.so (copy of an older build) vs csrc/libbev_mi355x.so
for rep in 1 2 3; do for lib in libbev_head.so libbev_mi355x.so; do
BEV_AMD_LIB=$PWD/point-cloud-preprocessing-tools_amd/csrc/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']))"
done; done
