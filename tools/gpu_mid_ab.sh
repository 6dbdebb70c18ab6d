#!/bin/bash
# round 4: the middle of the step (dense links in the chain, fused Dense(62) + softmax): parity, then same-box A/B of the bench step
python -m pytest tests/test_timit_parity.py tests/test_models.py tests/test_dp_gloo.py -m gpu -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "point or chain or head" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-extras --no-standalone --no-kernel-timing --steps 30"
for i in 1 2; do
  for env in "" "QK_NO_DENSE_IN_CHAIN=1" "QK_NO_FUSED_SOFTMAX=1" "QK_NO_DENSE_IN_CHAIN=1 QK_NO_FUSED_SOFTMAX=1"; do
    echo "[$env] $(env $env $B 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(round(r["ms_per_step"],3))')"
  done
done
