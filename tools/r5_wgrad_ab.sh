#!/bin/bash
# round 5: the LDS-DMA backward-weight band kernel against its round-2..4 form (QK_WGRAD_BAND_V1), same box, alternating
mkdir -p gpurun_out/r5
for r in 1 2; do
  for cf in "64 64" "32 32" "32 64"; do
    set -- $cf
    echo "== new  cq=$1 fq=$2"; python tools/power_trace.py --seconds 2.5 --kernel bwd_weight --cq $1 --fq $2
    echo "== v1   cq=$1 fq=$2"; QK_WGRAD_BAND_V1=1 python tools/power_trace.py --seconds 2.5 --kernel bwd_weight --cq $1 --fq $2
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/wgrad_ab.txt
