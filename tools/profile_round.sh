#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's rocprofv3 evidence in one go.
#   kernel-trace + stats of the default bench workload (BENCHMARK1), of the north-star grid with U3/C4 (ns512u3) and with
#   the stock schemes (ns512: HSIMT salinity), of BENCHMARK3 and of config 5 -- each also with the side streams off
#   (ROMS_HIP_OVERLAP=0: kernels run one at a time, so the per-kernel averages are the kernels' own durations) -- then the
#   FETCH_SIZE / WRITE_SIZE passes, and BENCHMARK1 with every periodic exchange through the mailbox (self-exchange).
# Usage: tools/profile_round.sh <round tag, e.g. r03>
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for spec in "benchmark1 20 b1" "ns512u3 10 ns512u3" "ns512 10 ns512" "benchmark3 8 b3" "config5 10 c5"; do
  set -- $spec
  bash tools/profile_gpu.sh $1 $2 ${R}_$3 | tail -1
  ROMS_HIP_OVERLAP=0 bash tools/profile_gpu.sh $1 $2 ${R}_$3_serial | tail -1
done
bash tools/profile_pmc.sh ns512u3 4 ${R}_ns512u3 | tail -2
bash tools/profile_pmc.sh benchmark1 6 ${R}_b1 | tail -2
bash tools/profile_pmc.sh benchmark3 3 ${R}_b3 | tail -2
bash tools/profile_pmc.sh config5 4 ${R}_c5 | tail -2
OUT=$ROOT/gpurun_out/prof_${R}_b1_selfx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  python3 $ROOT/tools/gpu_debug/gpu_selfx_prof.py benchmark1 20 peer > $OUT/trace.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv
grep selfx $OUT/trace.log
