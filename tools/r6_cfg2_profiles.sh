#!/bin/bash
# Round 6, BASELINE configs[1] (one QuaternionConv1D layer, fp32) after the K-contiguous rewrite of the fp32 kernels:
#   tools/grun.sh 2400 ./tools/r6_cfg2_profiles.sh          (then copy gpurun_out/r06c2/* into profiles/)
# 1. rocprofv3 --kernel-trace --stats of `bench.py --workload cfg2_qconv1d_timit_b64_fp32`   -> r06_cfg2_kernel_stats.csv
# 2. HBM traffic of the three kernels (tools/gpu_traffic.sh: FETCH_SIZE / WRITE_SIZE / request-size classes)  -> pmc_traffic_cfg2.json
# 3. SQ counters (MFMA-busy, instruction mix, LDS) of the three kernels                      -> r06_cfg2_<kernel>_pmc.txt
# 4. per-phase ablation (tools/ablate_cfg2.py)                                                -> r06_cfg2_ablation.txt
# 5. the default bench line (its cfg2_layer block) with the final library                     -> r06_bench_default_builder_run.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06c2; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ks_cfg2 --output-format csv -- python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-kernel-timing --workload cfg2_qconv1d_timit_b64_fp32 > $O/log_cfg2.txt 2>&1; echo "cfg2 trace rc=$?"
mv $O/ks_cfg2_kernel_stats.csv $O/r06_cfg2_kernel_stats.csv
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
rm -rf gpurun_out/traffic
KERNELS="fwd bwd_weight bwd_data" ./tools/gpu_traffic.sh cfg2_qconv1d_timit_b64_fp32 > $O/traffic_stdout.txt 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/pmc_traffic_cfg2.json
for K in fwd bwd_weight bwd_data; do
rm -rf gpurun_out/pmc
./tools/gpu_pmc.sh cfg2_qconv1d_timit_b64_fp32 $K "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU" > $O/r06_cfg2_${K}_pmc.txt 2>&1
done
python tools/ablate_cfg2.py 2>&1 | grep -v amdgpu.ids > $O/r06_cfg2_ablation.txt
( time python bench.py > $O/r06_bench_default_builder_run.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/bench_default.time | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06c2/r06_bench_default_builder_run.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['gpu_telemetry'].get('mean_sclk_mhz'))
c=d['cfg2_layer']; print(c['ms_per_step'], {k:(round(v['ms']*1e3,1), round(v['frac_of_peak'],3)) for k,v in c['kernels'].items()})
PY
tail -5 $O/traffic_stdout.txt
