#!/bin/bash
# round-2 iteration: GPU parity tests (all failures shown) + model bench + kernel trace of the model step
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
if [ "$1" != "notest" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r2/pytest_gpu.log
fi
for wl in cfg3_qcnn_timit_b256_bf16; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl > gpurun_out/r2/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
tail -1 gpurun_out/r2/bench_$wl.log | cut -c1-400
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_$wl -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --workload $wl > $GRAFT_REPO_ROOT/gpurun_out/r2/prof_$wl.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2/prof_$wl -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py $f > gpurun_out/r2/step_$wl.txt; tail -40 gpurun_out/r2/step_$wl.txt
done
