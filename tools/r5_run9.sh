python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py tests/test_timit_parity.py -q -m gpu -x -k "sixteen or 16to or sf16 or 48to16 or start_filter_16" 2>&1 | tail -30 > gpurun_out/r5_pytest9.txt
tail -30 gpurun_out/r5_pytest9.txt
python tools/ab_layers.py c16 c16to32 2>&1 | grep -v amdgpu | grep linear
QK_NO_SMALL16=1 python tools/ab_layers.py c16 c16to32 2>&1 | grep -v amdgpu | grep linear
