mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py tests/test_timit_parity.py tests/test_models.py -q -m gpu -k "point or head or dense or timit or chain or model or post" 2>&1 | tail -12 > gpurun_out/r5_pytest7.txt
tail -4 gpurun_out/r5_pytest7.txt
python tools/ab_layers.py head dense64 2>&1 | grep -v amdgpu
python bench.py --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_pt.json 2> gpurun_out/r5/bench_pt.err
python -c "
import json
d=json.load(open('gpurun_out/r5/bench_pt.json')); print('default', d['ms_per_step'], d['value'])
for c in d['in_step_kernels']['calls'][9:20]: print('  ', c['op'], c['rows'], c['n'], c['k'], c['calls_per_step'], round(c['ms'],4), round(c['frac_of_peak'],3), c['path'])
"
