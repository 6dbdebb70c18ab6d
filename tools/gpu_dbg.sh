#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_timit_parity.py tests/test_models.py -m gpu -q --no-header -p no:cacheprovider -k "fused_first_layer or timit_qcnn or bench_model_step or declines" > gpurun_out/dbg/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/dbg/pytest.txt | head -40
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-standalone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])
for c in d['in_step_kernels']['calls'][:18]: print(c['op'],c['rows'],c['n'],c['k'],c['path'],c['calls_per_step'],round(c['ms'],4),round(c['frac_of_peak'],3))"
