mkdir -p gpurun_out/c2
python tools/ablate_cfg2.py > gpurun_out/c2/tr1.txt 2>&1; cat gpurun_out/c2/tr1.txt | head -7
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py -x -q -m gpu -k "cfg2 or fp32 or f32" 2>&1 | tail -2
