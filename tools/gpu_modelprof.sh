#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/mp
WL=${1:-cfg3_qcnn_timit_b256_bf16}
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/mp -o ks --output-format csv -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-hamilton-gemm --workload $WL > gpurun_out/mp/log.txt 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/mp/ks_kernel_stats.csv')))
for r in rows[:22]:
    print('%-100s n=%5s avg=%9.1f us tot=%8.1f ms %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, float(r['Percentage'])))
PY
tail -1 gpurun_out/mp/log.txt | cut -c1-300
