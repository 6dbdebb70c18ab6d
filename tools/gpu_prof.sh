#!/bin/bash
# kernel-trace profile of the default bench (summary is copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/prof/bench_under_prof.log 2>&1
echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -30
find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -20 {}
