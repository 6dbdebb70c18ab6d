python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r5_pytest11.txt; tail -4 gpurun_out/r5_pytest11.txt
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_sf16_c.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r5/bench_sf16_c.json')); print('sf16', d['ms_per_step'], d['value'], d['qcnn_step']['frac_of_peak'])
for c in d['in_step_kernels']['calls'][:12]: print('  ', c['op'], c['rows'], c['n'], c['k'], c['calls_per_step'], round(c['ms'],4), round(c['frac_of_peak'],3), c['path'])
"
python bench.py --no-extras --no-cpu-baseline --no-standalone 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'])"
