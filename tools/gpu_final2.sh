#!/bin/bash
# round-2 evidence (copied into profiles/ afterwards): kernel-trace stats of the default bench (full QCNN step) and of
# the cfg2 layer, HBM traffic of every hot kernel (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ counters of
# the three Hamilton GEMM kernels of the 64 -> 64 body layer.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/final2
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final2 -o ks_qcnn --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-standalone > gpurun_out/final2/log_qcnn.txt 2>&1; echo "qcnn rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final2 -o ks_cfg2 --output-format csv -- python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-kernel-timing --workload cfg2_qconv1d_timit_b64_fp32 > gpurun_out/final2/log_cfg2.txt 2>&1; echo "cfg2 rc=$?"
python tools/trace_step.py gpurun_out/final2/ks_qcnn_kernel_trace.csv --all > gpurun_out/final2/qcnn_step_timeline.txt
rm -rf gpurun_out/traffic
./tools/gpu_traffic.sh cfg2_qconv1d_timit_b64_fp32 cfg3_body_qconv2d_b256_bf16 cfg3_stage1_qconv2d_b256_bf16 cfg3_32to64_qconv2d_b256_bf16 > gpurun_out/final2/traffic_stdout.txt 2>&1
cp gpurun_out/traffic/pmc_traffic.json gpurun_out/final2/
for K in fwd bwd_weight_chain bwd_data_chain; do
rm -rf gpurun_out/pmc
./tools/gpu_pmc.sh cfg3_body_qconv2d_b256_bf16 $K "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU" > gpurun_out/final2/pmc_cfg3body_$K.txt 2>&1
done
rm -f gpurun_out/final2/*agent_info.csv gpurun_out/final2/*domain_stats.csv
( time python bench.py > gpurun_out/final2/bench_default.json 2> gpurun_out/final2/bench_default.err ) 2> gpurun_out/final2/bench_default.time
python bench.py --workload cfg5_stack_b32_fp16 --no-cpu-baseline > gpurun_out/final2/bench_cfg5_stack.json 2>&1
ls gpurun_out/final2
