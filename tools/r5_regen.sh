STEPS="1 3 4 5 7 8 9" ./tools/reproduce_profiles.sh > gpurun_out/regen_log.txt 2>&1
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone > gpurun_out/r05/r05_bench_sf16_builder_run.json 2>/dev/null
tail -5 gpurun_out/regen_log.txt
