#!/bin/bash
# end-of-round evidence: kernel-trace stats for the three bench workloads + HBM traffic + SQ counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/final
for WL in cfg2_qconv1d_timit_b64_fp32 cfg3_body_qconv2d_b256_bf16 cfg3_qcnn_timit_b256_bf16; do
  ST=60; [ $WL = cfg3_qcnn_timit_b256_bf16 ] && ST=4
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final -o ks_$WL --output-format csv -- python bench.py --steps $ST --warmup 3 --no-cpu-baseline --no-hamilton-gemm --no-graph --workload $WL > gpurun_out/final/log_$WL.txt 2>&1
  echo "$WL rc=$?"
done
./tools/gpu_traffic.sh cfg2_qconv1d_timit_b64_fp32 cfg3_body_qconv2d_b256_bf16 > gpurun_out/final/traffic_stdout.txt 2>&1
cp gpurun_out/traffic/pmc_traffic.json gpurun_out/final/
for K in fwd bwd_weight bwd_data; do
./tools/gpu_pmc.sh cfg3_body_qconv2d_b256_bf16 $K "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU" > gpurun_out/final/pmc_cfg3body_$K.txt 2>&1
done
ls gpurun_out/final | head -40
