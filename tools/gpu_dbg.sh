#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
timeout 1500 python -m pytest tests/test_dp_gloo.py -m gpu -q --no-header -p no:cacheprovider -k "two_engine" > gpurun_out/dbg/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/dbg/pytest.txt | head -40
