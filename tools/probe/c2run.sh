python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py tests/test_models.py -x -q -m gpu 2>&1 | tail -2
