#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
for ab in 0 1 2; do
  echo "QK_ABLATE=$ab"
  QK_ABLATE=$ab timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (k, v['ms']*1e3) for k,v in d.get('kernels',{}).items()))
"
done
