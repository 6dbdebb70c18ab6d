mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py -q -m gpu -k "sixteen or 16to or sf16 or 48to16 or cq16" 2>&1 | tail -25 > gpurun_out/r5_pytest4.txt
tail -6 gpurun_out/r5_pytest4.txt
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_sf16.json 2> gpurun_out/r5/bench_sf16.err
QK_NO_BAND16=1 QK_NO_WGRAD_BAND=1 python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone --steps 20 --warmup 3 > gpurun_out/r5/bench_sf16_noband.json 2>> gpurun_out/r5/bench_sf16.err
tail -3 gpurun_out/r5/bench_sf16.err
python -c "
import json
for n in ('bench_sf16','bench_sf16_noband'):
    d=json.load(open('gpurun_out/r5/%s.json'%n)); print(n, d['ms_per_step'], d['value'], d.get('qcnn_step',{}).get('frac_of_peak'))
    for c in d['in_step_kernels']['calls'][:14]: print('  ', c['op'], c['rows'], c['n'], c['k'], c['calls_per_step'], round(c['ms'],4), round(c['frac_of_peak'],3), c['path'])
"
