#!/bin/bash
# PMC counters for one kernel: $1 workload, $2 kernel, rest = counters (one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pmc
WL=$1; K=$2; shift 2
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d gpurun_out/pmc -o pmc_${K}_$i --output-format csv -- python tools/run_kernel.py $WL $K 3 > gpurun_out/pmc/log_$i.txt 2>&1
  echo "pass $i ($grp) rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_hgemm' in k or 'k_wgrad' in k:
            agg[k[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(f.split('/')[-1], k)
        for c, v in d.items():
            print('    %-32s n=%d last=%.4g' % (c, len(v), v[-1]))
PY
