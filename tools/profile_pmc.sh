#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the two PMC passes the MI355X guide prescribes for HBM traffic
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains are combined with --pmc).
# Usage: tools/profile_pmc.sh <workload> <steps> <tag>     -> gpurun_out/pmc_<tag>/{fetch,write}/
set -u
WL=${1:-benchmark1}; STEPS=${2:-10}; TAG=${3:-b1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-breakdown --copy-probe"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass exit $?"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ARGS > $OUT/write.log 2>&1
echo "write pass exit $?"
grep -c . $OUT/fetch/*counter_collection.csv $OUT/write/*counter_collection.csv
