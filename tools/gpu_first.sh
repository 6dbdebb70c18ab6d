#!/bin/bash
# first GPU contact: smoke, parity tests, bench, kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -5 gpurun_out/smoke.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -5 gpurun_out/bench.log
