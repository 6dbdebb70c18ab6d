#!/bin/bash
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hamilton-gemm --workload cfg3_body_qconv2d_b256_fp32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (k, v['ms']*1e3) for k,v in d.get('kernels',{}).items()))
"
}
for sel in "" "0,16" "0,32" "1,8" "1,16" "4,16" "4,32" "2,16"; do
  echo "QK_BAND32=$sel"
  QK_BAND32=$sel run
done
