#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "native or channels_first" > gpurun_out/dbg/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/dbg/pytest.txt | head -40
for lay in channels_last native; do python bench.py --workload cfg3_body_qconv2d_b256_bf16 --layout $lay --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lay', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})"; done
