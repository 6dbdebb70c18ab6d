#!/bin/bash
# probe / A-B builds of the library: recompile qk_hgemm_bf16mfma.hip with extra -D flags and link it with the product's other objects:
#   tools/probe/build_variant.sh stamps -DQK_PHASE_STAMPS      -> tools/probe/libqk_stamps.so   (phase time stamps, tools/probe/phase_stamps.py)
#   (any other -D switch added to the kernel file for an A/B: load the result with QK_LIB=tools/probe/libqk_<name>.so)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../.."
P=quaternion-convolutional-neural-networks-for-end-to-end-automatic-speech-recognition_amd
mkdir -p /tmp/qk_variant_obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-value "$@" -c $P/csrc/qk_hgemm_bf16mfma.hip -o /tmp/qk_variant_obj/$NAME.o
OBJS=$(ls $P/csrc/_obj/*.o | grep -v qk_hgemm_bf16mfma.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/qk_variant_obj/$NAME.o -o tools/probe/libqk_$NAME.so
echo built tools/probe/libqk_$NAME.so
