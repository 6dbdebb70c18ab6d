#!/bin/bash
# parity tests that touch the 16-bit kernels (fast) -- tools/gpu_quick.sh [pytest -k expression]
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "${1:-half or 16bit or chain or conj or fold or timit or cfg5 or masked or full_size}" > gpurun_out/r2/pytest_quick.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2/pytest_quick.log
