#!/bin/bash
# A/B on one box, interleaved:  tools/gpu_ab.sh "<env assignments A>" "<env assignments B>" [shapes...]
#   e.g. tools/gpu_ab.sh "QK_LIB=$PWD/tools/libqk_old.bin" "" c64 c32      (old build vs current)
#        tools/gpu_ab.sh "QK_BAND16_8WAVES=1" "" c64                       (debug flag vs default)
A=$1; B=$2; shift; shift
mkdir -p gpurun_out/ab
for r in 1 2; do
env $A AB_TAG=A python tools/ab_layers.py "$@" 2>&1 | tee -a gpurun_out/ab/a.txt
env $B AB_TAG=B python tools/ab_layers.py "$@" 2>&1 | tee -a gpurun_out/ab/b.txt
done
