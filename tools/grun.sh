#!/bin/bash
# build libqk_hip.so + the C oracle here (hipcc cross-compiles), then run a command on the GPU box:
#   tools/grun.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
