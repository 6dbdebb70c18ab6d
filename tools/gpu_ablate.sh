#!/bin/bash
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hamilton-gemm "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step', round(d['ms_per_step'],4), ' '.join('%s %.1f (%.0f TF)' % (k, v['ms']*1e3, v['tflops']) for k,v in d.get('kernels',{}).items()))
"
}
for wl in "$@"; do echo $wl; run --workload $wl; done
