#!/bin/bash
# quick iteration: parity tests (fail-fast) + bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log
for wl in "$@"; do
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $wl > gpurun_out/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
tail -1 gpurun_out/bench_$wl.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],4))
for k,v in d.get('kernels',{}).items(): print(' ', k, round(v['ms']*1e3,1),'us', round(v['tflops'],1),'TF')
"
done
