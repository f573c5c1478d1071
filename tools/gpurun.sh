#!/bin/bash
# Build everything that travels to the GPU box (HIP library, Fortran host, oracle), then gpurun.
# usage: tools/gpurun.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "from roms_amd import build; build.build_hip(); build.build_host()" 2>&1 | grep -E "error|Error" && exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
