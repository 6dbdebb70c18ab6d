#!/bin/bash
# round 3, first GPU pass: the whole -m gpu suite, then the default bench (short) -- logs under gpurun_out/r3a
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r3a/pytest.txt 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3a/pytest.txt | tail -40
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['config']['workload'])
for k in ('qcnn_prelu_dropout_step','qcnn_nodropout_step','qcnn_ctc_step'):
    print(k, d.get(k))
print('roofline', {k:v for k,v in d.get('roofline',{}).items() if k!='standalone'})
for c in d.get('in_step_kernels',{}).get('calls',[]): print(c['op'],c['rows'],c['n'],c['k'],c['path'],c['calls_per_step'],round(c['ms'],4),round(c['frac_of_peak'],3))
PY
