#!/bin/bash
# probe: PMC counters of the fused first-layer kernels (tools/probe/c1_time.py as the workload), one rocprofv3 pass per group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/c1pmc
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d gpurun_out/c1pmc -o c1_$i --output-format csv -- python tools/probe/c1_time.py > gpurun_out/c1pmc/log_$i.txt 2>&1
  echo "pass $i ($grp) rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/c1pmc/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_conv1_pool' in k:
            agg['k_conv1_pool_fwd' if 'pool_fwd' in k else 'k_conv1_pool_bwd'][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(agg.items()):
        for c, v in d.items():
            print('%-18s %-28s launches=%d median=%.4g' % (k, c, len(v), sorted(v)[len(v)//2]))
PY
