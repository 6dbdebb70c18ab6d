python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py -x -q -m gpu -k "cfg2 or fp32 or f32" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-standalone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', d['ms_per_step']); c=d['cfg2_layer']; print(c['ms_per_step'], c.get('sustained_clock_mhz'), c.get('mean_socket_w'), {k:(round(v['ms']*1e3,1), round(v['frac_of_peak'],3), round(v.get('frac_at_sustained_clock',0),3)) for k,v in c['kernels'].items()})"
