mkdir -p gpurun_out/c2
python tools/ablate_cfg2.py > gpurun_out/c2/wg7.txt 2>&1; cat gpurun_out/c2/wg7.txt | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
