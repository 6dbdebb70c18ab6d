#!/bin/bash
# Regenerates every profiles/<round>_* file from a clean GPU lease (one MI355X box); RTAG picks the round prefix (default r05):
#     tools/grun.sh 3000 ./tools/reproduce_profiles.sh        (then: cp gpurun_out/r05/* profiles/)
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench (full TIMIT QCNN step: relu + dropout + l2, CTC cost, bf16, B = 256)
#    + the last step as a kernel-by-kernel timeline;                       -> ${RTAG}_qcnn_bf16_b256_kernel_stats.csv, ${RTAG}_qcnn_step_timeline.txt
# 2. the same for BASELINE configs[1] (one QuaternionConv1D layer, fp32);   -> ${RTAG}_cfg2_kernel_stats.csv
# 3. HBM traffic of every hot kernel (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, gfx950 x2 on FETCH_SIZE);  -> pmc_traffic.json
# 4. SQ counters of the three kernels of the 64 -> 64 layer as the step launches them;       -> ${RTAG}_cfg3body_bf16_*_pmc.txt
# 5. power / shader-clock traces on different operand values (tools/power_trace.py);           -> ${RTAG}_power_clock_trace_*.txt
# 6. the energy-apportioning probe (tools/probe: MFMA only / + LDS reads / + staging, 32- and 64-row wave tiles); -> ${RTAG}_energy_probe.txt
# 7. per-phase ablation of the 16-bit kernels (tools/ablate16.py);                             -> ${RTAG}_ablation.txt
# 8. the bench lines: default (CTC), --loss sum, cfg5 stack, native-layout layer (builder runs).  -> ${RTAG}_bench_*.json
# 9. phase time stamps of the band kernels (probe build, tools/probe/phase_stamps.py) and the sum / CTC loss A-B with telemetry (ab_loss.py)  -> ${RTAG}_phase_stamps.txt, ${RTAG}_loss_ab.txt
# 10. the fused first layer (k_conv1_pool_fwd / _bwd at B = 256): stand-alone times and SQ counters (tools/probe/c1_time.py, c1_pmc.sh)  -> ${RTAG}_first_layer_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; RTAG=${RTAG:-r05}; O=gpurun_out/$RTAG; rm -rf $O; mkdir -p $O
STEPS="${STEPS:-1 2 3 4 5 6 7 8 9 10}"
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has 1; then
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ks_qcnn --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-standalone > $O/log_qcnn.txt 2>&1; echo "qcnn trace rc=$?"
python tools/trace_step.py $O/ks_qcnn_kernel_trace.csv --all > $O/${RTAG}_qcnn_step_timeline.txt
mv $O/ks_qcnn_kernel_stats.csv $O/${RTAG}_qcnn_bf16_b256_kernel_stats.csv
fi
if has 2; then
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ks_cfg2 --output-format csv -- python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-kernel-timing --workload cfg2_qconv1d_timit_b64_fp32 > $O/log_cfg2.txt 2>&1; echo "cfg2 trace rc=$?"
mv $O/ks_cfg2_kernel_stats.csv $O/${RTAG}_cfg2_kernel_stats.csv
fi
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
if has 3; then
rm -rf gpurun_out/traffic
./tools/gpu_traffic.sh cfg2_qconv1d_timit_b64_fp32 cfg3_body_qconv2d_b256_bf16 cfg3_stage1_qconv2d_b256_bf16 cfg3_32to64_qconv2d_b256_bf16 > $O/traffic_stdout.txt 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/
fi
if has 4; then
for K in fwd bwd_weight_chain bwd_data_chain; do
rm -rf gpurun_out/pmc
./tools/gpu_pmc.sh cfg3_body_qconv2d_b256_bf16 $K "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU" > $O/${RTAG}_cfg3body_bf16_${K}_pmc.txt 2>&1
done
fi
if has 5; then
for K in fwd bwd_data bwd_weight; do
python tools/power_trace.py --seconds 3 --kernel $K 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_power_clock_trace_$K.txt
python tools/power_trace.py --seconds 3 --kernel $K --cq 32 --fq 32 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_power_clock_trace_32to32_$K.txt
done
fi
if has 6; then
python tools/probe/energy_probe.py --seconds 3 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_energy_probe.txt
fi
if has 7; then
python tools/ablate16.py c64 c32 c32to64 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_ablation.txt
fi
if has 8; then
( time python bench.py > $O/${RTAG}_bench_default_builder_run.json 2> $O/bench_default.err ) 2> $O/bench_default.time
python bench.py --loss sum --no-extras --no-standalone --no-cpu-baseline > $O/${RTAG}_bench_sumloss_builder_run.json 2>/dev/null
python bench.py --workload cfg5_stack_b32_fp16 --no-cpu-baseline > $O/${RTAG}_bench_cfg5_stack_builder_run.json 2>/dev/null
for lay in channels_last native; do python bench.py --workload cfg3_body_qconv2d_b256_bf16 --layout $lay --no-cpu-baseline > $O/${RTAG}_bench_cfg3body_${lay}_builder_run.json 2>/dev/null; done
QK_DP_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" > $O/${RTAG}_bench_one_rank_rccl_builder_run.json
cat $O/bench_default.time
fi
if has 9; then
QK_LIB=$R/tools/probe/libqk_stamps.so python tools/probe/phase_stamps.py c64 c32 c32to64 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_phase_stamps.txt
python tools/probe/ab_loss.py 2>&1 | grep -v amdgpu.ids > $O/${RTAG}_loss_ab.txt
python tools/probe/grad_stats.py 2>&1 | grep -v amdgpu.ids >> $O/${RTAG}_loss_ab.txt
fi
ls -la $O
if has 10; then
( python tools/probe/c1_time.py 2>&1 | grep -v amdgpu.ids
  ./tools/probe/c1_pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" 2>&1 ) > $O/${RTAG}_first_layer_pmc.txt
fi
if has 11; then
# round 5: the LDS-DMA backward-weight kernel against its round-2..4 form, same box, alternating; the sf = 16 model; graph replay vs eager
bash tools/r5_wgrad_ab.sh > /dev/null 2>&1; cp gpurun_out/r5/wgrad_ab.txt $O/${RTAG}_wgrad_ab.txt
python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --no-extras --no-cpu-baseline --no-standalone > $O/${RTAG}_bench_sf16_builder_run.json 2>/dev/null
for i in 1 2; do
python bench.py --no-extras --no-cpu-baseline --no-standalone 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['ms_per_step'], d['gpu_telemetry']['mean_sclk_mhz'])"
python bench.py --graph --no-extras --no-cpu-baseline --no-standalone 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], d['gpu_telemetry']['mean_sclk_mhz'])"
done > $O/${RTAG}_graph_vs_eager.txt
fi
