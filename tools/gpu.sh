#!/bin/bash
# Rebuild the library HERE (the .so travels with the snapshot) and run a command on the GPU box:
#   tools/gpu.sh [timeout_s] '<command>'
set -e
cd "$(dirname "$0")/.."
t=900
if [[ "$1" =~ ^[0-9]+$ ]]; then t=$1; shift; fi
make -s -C unflow_amd/csrc -j4 2>&1 | grep -E "error|Error" && exit 1
make -s -C oracle >/dev/null
exec /usr/local/graft/bin/gpurun --timeout $t -- "$1"
