#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/power

for k in fwd bwd_data bwd_weight; do python tools/power_trace.py --seconds 3 --kernel $k 2>&1 | grep -v amdgpu.ids | tee gpurun_out/power/trace_$k.txt; done

