mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py -q -m gpu -k "sixteen or 16to or sf16 or 48to16 or cq16 or 3tap or 32to64 or body" 2>&1 | tail -12 > gpurun_out/r5_pytest6.txt
tail -4 gpurun_out/r5_pytest6.txt
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_sf16_b.json 2> gpurun_out/r5/bench_sf16_b.err
python -c "
import json
d=json.load(open('gpurun_out/r5/bench_sf16_b.json')); print('sf16', d['ms_per_step'], d['value'], d.get('qcnn_step',{}).get('frac_of_peak'))
for c in d['in_step_kernels']['calls'][:16]: print('  ', c['op'], c['rows'], c['n'], c['k'], c['calls_per_step'], round(c['ms'],4), round(c['frac_of_peak'],3), c['path'])
"
echo "== 8-wave band forms vs 4-wave (A = QK_BAND16_8WAVES)"; bash tools/gpu_ab.sh "QK_BAND16_8WAVES=1" "" c64 c32 2>&1 | grep -v amdgpu | grep "fwd\|bwd_data" | grep linear
echo "== phase stamps"; QK_LIB=$PWD/tools/probe/libqk_stamps.so python tools/probe/phase_stamps.py c64 c32 2>&1 | grep -v amdgpu.ids | tail -30
