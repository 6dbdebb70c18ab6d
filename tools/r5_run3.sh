set -x
mkdir -p gpurun_out/r5
QK_DP_SHARE_DEVICE=1 QK_DP_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/r5/b8.out 2> gpurun_out/r5/b8.err; echo "rc $?" >> gpurun_out/r5/b8.out
grep -n "Error\|error\|fault\|Traceback" gpurun_out/r5/b8.err | head -20
python -m pytest tests -q -m gpu --deselect "tests/test_dp_gloo.py::test_plain_bench_gpus_n_starts_its_ranks_and_prints_one_line[8]" 2>&1 | tail -25 > gpurun_out/r5_pytest3.txt
tail -8 gpurun_out/r5_pytest3.txt
python bench.py --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_eager.json 2> gpurun_out/r5/bench_eager.err
python bench.py --graph --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_graph.json 2> gpurun_out/r5/bench_graph.err
python bench.py --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_eager2.json 2>> gpurun_out/r5/bench_eager.err
python bench.py --graph --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r5/bench_graph2.json 2>> gpurun_out/r5/bench_graph.err
tail -3 gpurun_out/r5/bench_graph.err
python -c "
import json
for n in ('bench_eager','bench_graph','bench_eager2','bench_graph2'):
    try:
        d=json.load(open('gpurun_out/r5/%s.json'%n)); print(n, d['ms_per_step'], d['config']['launch'], d['gpu_telemetry']['mean_sclk_mhz'])
    except Exception as e: print(n, 'ERR', e)
"
