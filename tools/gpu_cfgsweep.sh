#!/bin/bash
export TMPDIR=/tmp
for cfg in "" "2,8" "2,16" "2,32" "5,8" "5,32" "6,16" "3,8" "3,32"; do
  echo "QK_FORCE_CFG=$cfg"
  QK_FORCE_CFG=$cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-hamilton-gemm 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (k, v['ms']*1e3) for k,v in d.get('kernels',{}).items()))
"
done
