#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of one bench.py workload.
# Usage: tools/profile_gpu.sh <workload> <steps> <tag>
# Results land under gpurun_out/prof_<tag>/; copy the *_kernel_stats.csv into profiles/.
# (PMC passes are not part of this script: see DESIGN.md 6.4.)
set -u
WL=${1:-benchmark1}; STEPS=${2:-20}; TAG=${3:-b1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-breakdown > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_line.json
rm -f $OUT/trace/*kernel_trace.csv
ls $OUT/trace
