for w in os1_firing oxford_concat; do
F=1000; if [ $w = oxford_concat ]; then F=100; fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --workload $w --frames $F 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); F=d['config']['frames_per_gpu']
print('$w', round(d['value']), 'B_frame', round(d['config']['algorithmic_bytes_per_frame']/1e6,2), 'MB', 'pipeline frac', round(d['roofline']['pipeline']['frac'],3), [(k['name'][2:8], round(k['avg_launch_ms']*1e3*k['launches']/d['steps']/F,2)) for k in d['kernels']])"
done
