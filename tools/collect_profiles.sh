#!/bin/bash
# gpurun_out/prof_<R>_* and pmc_<R>_* (tools/profile_round.sh) -> profiles/<R>_*: kernel stats, bench lines, traffic JSONs.
# The byte count of the calibration kernel of a workload (k_copy_probe: nij*(N+1)*8 bytes read, as many written) is taken
# from the round-3 file of the same workload (same grids).      usage: tools/collect_profiles.sh r04
R=${1:-r04}
cd "$(dirname "$0")/.."
for d in gpurun_out/prof_${R}_*; do
  tag=${d#gpurun_out/prof_}
  [ -f $d/trace/t_kernel_stats.csv ] && cp $d/trace/t_kernel_stats.csv profiles/${tag}_kernel_stats.csv
  [ -s $d/bench_line.json ] && cp $d/bench_line.json profiles/${tag}_bench_line.json
done
[ -f profiles/${R}_b1_selfx_kernel_stats.csv ] && mv profiles/${R}_b1_selfx_kernel_stats.csv profiles/${R}_b1_selfx_mailbox_kernel_stats.csv
for spec in "ns512u3 ns512u3" "b1 benchmark1" "b3 benchmark3" "c5 config5"; do
  set -- $spec
  [ -d gpurun_out/pmc_${R}_$1 ] || continue
  cb=$(python3 -c "import json; print(json.load(open('profiles/r03_$1_traffic.json'))['calibration']['bytes_read_per_launch'])")
  python3 tools/summarize_pmc.py $R ${R}_$1 $2 $cb
  mv profiles/${R}_${R}_$1_traffic.json profiles/${R}_$1_traffic.json
done
