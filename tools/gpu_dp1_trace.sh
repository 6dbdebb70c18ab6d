#!/bin/bash
# kernel timeline of the one-rank RCCL bench step: where do the extra microseconds per step go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/dp1
QK_DP_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 600 rocprofv3 --kernel-trace -d gpurun_out/dp1 -o tr --output-format csv -- python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-hamilton-gemm --no-kernel-timing > gpurun_out/dp1/log.txt 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/dp1/tr_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 6 steps worth of kernels
names = [r['Kernel_Name'][:40] for r in rows]
tail = rows[-48:]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = None
for r in tail:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = (s - prev_end) if prev_end is not None else 0
    print('%8.1f %8.1f dur %6.1f gap %6.1f q%s %s' % (s/1e3, e/1e3, (e-s)/1e3, gap/1e3, r.get('Queue_Id','?'), r['Kernel_Name'][:60]))
    prev_end = max(prev_end or 0, e)
PY
