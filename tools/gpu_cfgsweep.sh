#!/bin/bash
export TMPDIR=/tmp
for cfg in "" "0,8" "0,16" "0,32" "0,40" "5,8" "5,32" "5,40" "6,8" "6,16" "2,8" "2,32" "3,32" "4,32"; do
  echo "QK_FORCE_CFG=$cfg"
  QK_FORCE_CFG=$cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (k, v['ms']*1e3) for k,v in d.get('kernels',{}).items()))
"
done
