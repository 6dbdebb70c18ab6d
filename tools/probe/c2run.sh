python -m pytest tests/test_fullsize_oracle_parity.py -x -q -m gpu -k "fp32_layer_on_the_big" -s 2>&1 | tail -8
