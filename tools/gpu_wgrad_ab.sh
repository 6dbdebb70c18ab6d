#!/bin/bash
# round 4: backward-weight band kernel after a change -- parity first, then stand-alone timing on dense data and the power trace
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py tests/test_timit_parity.py -m gpu -x -q 2>&1 | tail -3
python tools/ab_layers.py c64 c32 c32to64 2>/dev/null | grep "bwd_weight"
python tools/power_trace.py --kernel bwd_weight --seconds 2.5 2>/dev/null
