python tools/ablate_point.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_timit_parity.py -q -m gpu -k "start_filter_16" 2>&1 | tail -3
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sf16', d['ms_per_step'], d['value'], d['qcnn_step']['frac_of_peak'])
for c in d['in_step_kernels']['calls'][:8]: print('  ', c['op'], c['rows'], c['n'], c['k'], c['calls_per_step'], round(c['ms'],4), round(c['frac_of_peak'],3), c['path'])
"
