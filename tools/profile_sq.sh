#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ / TA counter passes of one bench.py workload (no trace domains
# combined with --pmc).  Usage: tools/profile_sq.sh <workload> <steps> <tag>  -> gpurun_out/sq_<tag>/
set -u
WL=${1:-ns512u3}; STEPS=${2:-2}; TAG=${3:-ns512u3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-breakdown"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -o p -- python3 $ARGS > $OUT/p1.log 2>&1
echo "pass1 exit $?"
# (a second pass with GRBM_GUI_ACTIVE TA_TA_BUSY TA_*_STALLED_BY_TC_CYCLES MemUnitStalled OccupancyPercent
#  hung rocprofv3 on this pool until the time-out: not collected)
