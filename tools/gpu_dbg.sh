#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/ctc; rm -rf $O; mkdir -p $O
for l in ctc sum; do
timeout 900 rocprofv3 --kernel-trace -d $O -o ks_$l --output-format csv -- python bench.py --loss $l --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extras --no-standalone > $O/log_$l.txt 2>&1; echo "rc=$?"
python tools/trace_step.py $O/ks_${l}_kernel_trace.csv --all > $O/timeline_$l.txt
grep "^kernels" $O/timeline_$l.txt
python - <<PY
rows=[]
for l in open('$O/timeline_$l.txt'):
    p=l.split(None,2)
    try: rows.append((float(p[0]),float(p[1]),p[2].strip()[:60]))
    except Exception: pass
prev=None; tot=0
for s,d,n in rows:
    if prev is not None and s-prev>8: print('  gap %.1f us before %s (at %.0f)'%(s-prev,n,s)); tot+=s-prev
    prev=s+d
print('  total gaps', tot)
PY
rm -f $O/ks_${l}_kernel_trace.csv
done
