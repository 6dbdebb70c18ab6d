#!/bin/bash
# round 5: the band kernel's in-place fragment pipeline (QK_BAND_PIPE, default) against the same kernel without it
# (tools/probe/build_variant.sh nopipe -DQK_BAND_PIPE=0), same box, alternating
mkdir -p gpurun_out/r5
NP=$PWD/tools/probe/libqk_nopipe.so
for r in 1 2; do
  for cf in "64 64" "32 32" "32 64"; do
    set -- $cf
    for k in fwd bwd_data; do
      echo "== pipe    $k cq=$1 fq=$2"; python tools/power_trace.py --seconds 2 --kernel $k --cq $1 --fq $2
      echo "== nopipe  $k cq=$1 fq=$2"; QK_LIB=$NP python tools/power_trace.py --seconds 2 --kernel $k --cq $1 --fq $2
    done
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/pipe_ab.txt
