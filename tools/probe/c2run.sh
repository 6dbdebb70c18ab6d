python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
mkdir -p gpurun_out/final
( time python bench.py > gpurun_out/final/r06_bench_default_builder_run.json 2> gpurun_out/final/bench.err ) 2> gpurun_out/final/bench.time; tail -3 gpurun_out/final/bench.time
python bench.py --workload cfg2_qconv1d_timit_b64_fp32 --no-cpu-baseline > gpurun_out/final/r06_bench_cfg2_builder_run.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/r06_bench_default_builder_run.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['gpu_telemetry'].get('mean_sclk_mhz'), d['gpu_telemetry'].get('mean_socket_w'), d.get('energy_j_per_step'))
c=d['cfg2_layer']; print(c['ms_per_step'], c.get('sustained_clock_mhz'), {k:(round(v['ms']*1e3,1), round(v['frac_of_peak'],3)) for k,v in c['kernels'].items()})
print('cfg5', d['cfg5_stack']['ms_per_step'], 'sf16', d['qcnn_sf16_step'].get('ms_per_step'), 'prelu', d['qcnn_prelu_dropout_step'].get('ms_per_step'))
print('cpu', d['cpu_baseline']['value'], d['roofline']['frac'])
PY
