python tools/ablate_cfg2.py 2>&1 | grep -v amdgpu | head -12
python -m pytest tests/test_gpu_parity.py tests/test_fullsize_oracle_parity.py -x -q -m gpu -k "cfg2 or fp32 or f32" 2>&1 | tail -2
python bench.py --no-cpu-baseline --workload cfg3_body_qconv2d_b256_fp32 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg3 fp32 body step', round(d['ms_per_step'],2), {k:(round(v['ms'],3), round(v['frac_of_peak'],3)) for k,v in d['kernels'].items()})"
