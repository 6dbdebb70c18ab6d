#!/bin/bash
# what holds a LONE workgroup of the band kernel below the MFMA rate?  probe build (QK_BAND_PROBE), zeros, QK_ABLATE bits:
#   64 = one workgroup per CU, 16 = no waits / barriers in the K loop, 32 = no DMA issued in the K loop, 8 = no epilogue
mkdir -p gpurun_out/r5
L=$PWD/tools/probe/libqk_probe.so
for cf in "64 64" "32 32"; do
  set -- $cf
  for ab in 0 16 32 48 64 80 96 112 72 120; do
    echo "== ablate $ab cq=$1 fq=$2"; QK_ABLATE=$ab QK_LIB=$L python tools/power_trace.py --seconds 0.6 --kernel fwd --cq $1 --fq $2 | grep -E "zeros"
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/band_lone.txt
