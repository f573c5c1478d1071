#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats and the two PMC passes the
# MI355X guide prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE do not fit one pass), for one
# bench.py workload.  Usage: tools/profile_gpu.sh <workload> <steps> <tag>
# Results land under gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into profiles/.
set -u
WL=${1:-benchmark1}; STEPS=${2:-20}; TAG=${3:-b1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-breakdown --copy-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $ARGS > $OUT/write.log 2>&1
ls -R $OUT | head -30
# kernel_trace.csv of the PMC passes is not needed
rm -f $OUT/fetch/*kernel_trace.csv $OUT/write/*kernel_trace.csv $OUT/trace/*kernel_trace.csv
