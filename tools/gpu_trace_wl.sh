#!/bin/bash
# kernel timeline of one training step of a bench workload: tools/gpu_trace_wl.sh <workload> [extra bench args]
WL=$1; shift
mkdir -p gpurun_out/r2; cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2/trace_$WL
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/trace_$WL -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --workload $WL "$@" > $GRAFT_REPO_ROOT/gpurun_out/r2/trace_$WL.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $(find gpurun_out/r2/trace_$WL -name '*kernel_trace.csv' | head -1) > gpurun_out/r2/step_$WL.txt; tail -60 gpurun_out/r2/step_$WL.txt
