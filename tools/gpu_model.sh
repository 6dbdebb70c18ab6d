#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_models.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
for wl in "$@"; do
timeout 900 python bench.py --steps 5 --warmup 2 --workload $wl > gpurun_out/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
tail -3 gpurun_out/bench_$wl.log | cut -c1-900
done
