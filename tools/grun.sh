#!/bin/bash
# build libqk_hip.so + the C oracle here (hipcc cross-compiles), then run a command on the GPU box:
#   tools/grun.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
[ tools/probe/libenergy_probe.so -nt tools/probe/energy_probe.hip ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe/energy_probe.hip -o tools/probe/libenergy_probe.so
[ tools/probe/liboperand_probe.so -nt tools/probe/operand_probe.hip ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe/operand_probe.hip -o tools/probe/liboperand_probe.so
[ tools/probe/libqk_stamps.so -nt quaternion-convolutional-neural-networks-for-end-to-end-automatic-speech-recognition_amd/libqk_hip.so ] || bash tools/probe/build_variant.sh stamps -DQK_PHASE_STAMPS | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
