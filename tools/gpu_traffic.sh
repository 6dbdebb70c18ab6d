#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) for each hot kernel of a workload.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/traffic
for WL in "$@"; do
for K in fwd bwd_weight bwd_data bwd_weight_chain bwd_data_chain; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C -d gpurun_out/traffic -o ${WL}_${K}_${C} --output-format csv -- python tools/run_kernel.py $WL $K 3 > gpurun_out/traffic/log.txt 2>&1 || echo "fail $WL $K $C"
  done
done
done
python - <<'PY'
import csv, glob, json, os, re, collections
out = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/traffic/*_counter_collection.csv')):
    m = re.match(r'(.*?)_(fwd|bwd_weight_chain|bwd_data_chain|bwd_weight|bwd_data)_(FETCH_SIZE|WRITE_SIZE)_counter_collection.csv', os.path.basename(f))
    if not m: continue
    wl, k, c = m.groups()
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if any(s in name for s in ('k_hgemm', 'k_wgrad')):
            vals[name].append(float(r['Counter_Value']))
    # the kernel under test is the one launched most often in this run (run_kernel.py: 3 reps + 1 fwd warm-up)
    want = 'k_wgrad' if 'weight' in k else 'k_hgemm'
    cands = {n: v for n, v in vals.items() if want in n}
    if k == 'bwd_data' and len(cands) > 1:      # the warm-up fwd is also a k_hgemm: take the most frequent
        pass
    name = max(cands, key=lambda n: len(cands[n]))
    out[wl].setdefault(k, {})[c] = cands[name][-1]
    out[wl][k]['kernel'] = name[:120]
for wl in out:
    for k, d in out[wl].items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
            d['hbm_bytes'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
json.dump(out, open('gpurun_out/traffic/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
