#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "half_matches or full_size or masked_dy or chain or accumulating or cfg5" > gpurun_out/dbg/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/dbg/pytest.txt | head -40
for k in bwd_weight; do python tools/power_trace.py --seconds 2 --kernel $k 2>&1 | grep -v amdgpu.ids; python tools/power_trace.py --seconds 2 --kernel $k --cq 32 --fq 32 2>&1 | grep -v amdgpu.ids;  python tools/power_trace.py --seconds 2 --kernel $k --cq 32 --fq 64 2>&1 | grep -v amdgpu.ids; done
