#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_timit_parity.py tests/test_dp_gloo.py -m gpu -q --no-header -p no:cacheprovider -k "relu_dropout or declines or full_size_properties or homogeneity or 16bit_matches or l2_regulariser or fused_first_layer" > gpurun_out/dbg/pytest.txt 2>&1; echo rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/dbg/pytest.txt | head -60
