#!/bin/bash
# HBM traffic of every hot kernel of the given layer workloads (bench.WORKLOADS, kind 'conv'), separate rocprofv3 --pmc passes:
#   pass 1  FETCH_SIZE                          (derived; gfx94x formula: 128-byte requests tallied at 64 -> the guide's x2 for wide streaming reads)
#   pass 2  WRITE_SIZE
#   pass 3  TCC_EA0_RDREQ_sum + _32B + _64B + _128B    (round 6: the request-size classes themselves: read bytes = 32 n32 + 64 n64 + 128 n128 --
#                                                       settles whether the x2 applies to a kernel's access pattern instead of assuming it)
#   pass 4  TCC_EA0_WRREQ_sum + _64B
# plus the fused first layer (k_conv1_pool_fwd / _bwd, tools/probe/c1_time.py) when FIRST=1.
#   tools/gpu_traffic.sh <workload> ...        [KERNELS="fwd bwd_weight_chain bwd_data_chain"] [FIRST=1]
# -> gpurun_out/traffic/pmc_traffic.json: per workload / kernel  FETCH_SIZE, WRITE_SIZE (KiB), the request counts, and
#      hbm_bytes        = (2 FETCH_SIZE + WRITE_SIZE) KiB   -- the guide's recipe (MI355X_MICROARCH.md, HBM): what bench.py reports as `traffic`
#      hbm_bytes_exact  = 32 n32 + 64 n64 + 128 n128 + write bytes  -- from the size classes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/traffic
KERNELS="${KERNELS:-fwd bwd_weight bwd_data bwd_weight_chain bwd_data_chain}"
P1="FETCH_SIZE"; P2="WRITE_SIZE"
P3="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P4="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
for WL in "$@"; do
for K in $KERNELS; do
  i=0
  for C in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $C -d gpurun_out/traffic -o ${WL}__${K}__p$i --output-format csv -- python tools/run_kernel.py $WL $K 3 > gpurun_out/traffic/log.txt 2>&1 || echo "fail $WL $K pass $i"
  done
done
done
if [ "$FIRST" = "1" ]; then
  i=0
  for C in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $C -d gpurun_out/traffic -o first_layer__fwdbwd__p$i --output-format csv -- python tools/probe/c1_time.py > gpurun_out/traffic/log.txt 2>&1 || echo "fail first layer pass $i"
  done
fi
python - <<'PY'
import csv, glob, json, os, re, collections
FAMS = ('k_hgemm', 'k_wgrad', 'k_hconv16_small', 'k_conv1_pool')
raw = collections.defaultdict(lambda: collections.defaultdict(dict))      # wl -> kernel -> counter -> value
names = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/traffic/*_counter_collection.csv')):
    m = re.match(r'(.*?)__(.*?)__p\d+_counter_collection.csv', os.path.basename(f))
    if not m:
        continue
    wl, k = m.groups()
    vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel name -> counter -> values
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if any(s in n for s in FAMS):
            vals[n][r['Counter_Name']].append(float(r['Counter_Value']))
    if wl == 'first_layer':
        for tag in ('k_conv1_pool_fwd', 'k_conv1_pool_bwd'):
            for n, d in vals.items():
                if tag in n:
                    for c, v in d.items():
                        raw[wl][tag][c] = sorted(v)[len(v) // 2]
                    names[wl][tag] = n[:120]
        continue
    # the kernel under test is the one launched most often in the run (run_kernel.py: 3 reps + 1 forward warm-up)
    want = ('k_wgrad',) if 'weight' in k else ('k_hgemm', 'k_hconv16_small')
    cands = {n: d for n, d in vals.items() if any(w in n for w in want)}
    if not cands:
        continue
    n = max(cands, key=lambda q: max(len(v) for v in cands[q].values()))
    for c, v in cands[n].items():
        raw[wl][k][c] = v[-1]
    names[wl][k] = n[:120]
out = {}
for wl in raw:
    out[wl] = {}
    for k, d in raw[wl].items():
        e = dict(d)
        e['kernel'] = names[wl][k]
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
            e['hbm_bytes'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
        if 'TCC_EA0_RDREQ_sum' in d:
            n32, n64, n128 = d.get('TCC_EA0_RDREQ_32B_sum', 0.0), d.get('TCC_EA0_RDREQ_64B_sum', 0.0), d.get('TCC_EA0_RDREQ_128B_sum', 0.0)
            e['read_bytes_exact'] = 32 * n32 + 64 * n64 + 128 * n128
            e['rdreq_unclassified'] = d['TCC_EA0_RDREQ_sum'] - n32 - n64 - n128
            if 'FETCH_SIZE' in d and d['FETCH_SIZE'] > 0:
                e['fetch_size_correction'] = e['read_bytes_exact'] / (d['FETCH_SIZE'] * 1024)      # what FETCH_SIZE has to be multiplied by for THIS kernel
        if 'TCC_EA0_WRREQ_sum' in d:
            w64 = d.get('TCC_EA0_WRREQ_64B_sum', 0.0)
            e['write_bytes_exact'] = 64 * w64 + 32 * (d['TCC_EA0_WRREQ_sum'] - w64)
        if 'read_bytes_exact' in e and 'write_bytes_exact' in e:
            e['hbm_bytes_exact'] = e['read_bytes_exact'] + e['write_bytes_exact']
        out[wl][k] = e
json.dump(out, open('gpurun_out/traffic/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for wl in sorted(out):
    for k in sorted(out[wl]):
        e = out[wl][k]
        print('%-34s %-18s guide %8.1f MB  exact %8.1f MB  (reads: %.0f x32 %.0f x64 %.0f x128; FETCH_SIZE x %.2f)  %s' % (
            wl, k, e.get('hbm_bytes', 0) / 1e6, e.get('hbm_bytes_exact', 0) / 1e6, e.get('TCC_EA0_RDREQ_32B_sum', 0), e.get('TCC_EA0_RDREQ_64B_sum', 0),
            e.get('TCC_EA0_RDREQ_128B_sum', 0), e.get('fetch_size_correction', 0), e['kernel'][40:100]))
PY
