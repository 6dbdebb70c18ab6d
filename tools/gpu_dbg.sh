#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/dp1; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $O/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.txt | head -20
timeout 900 rocprofv3 --kernel-trace -d $O -o ks --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/log.txt 2>&1; echo "rc=$?"
python tools/trace_step.py $O/ks_kernel_trace.csv --all > $O/timeline.txt
tail -22 $O/timeline.txt | cut -c1-110
rm -f $O/ks_kernel_trace.csv
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-standalone --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
