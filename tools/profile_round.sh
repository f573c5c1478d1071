#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's rocprofv3 evidence in one go.
#   kernel-trace + stats of the default bench workload (BENCHMARK1) and of the north-star grid (ns512u3),
#   the latter also with the side stream off (ROMS_HIP_OVERLAP=0: kernels run one at a time, so the per-kernel
#   averages are comparable with bench.py's synchronous per-kernel pass), and the FETCH_SIZE / WRITE_SIZE passes.
# Usage: tools/profile_round.sh <round tag, e.g. r02>
R=${1:-r02}
bash tools/profile_gpu.sh benchmark1 20 ${R}_b1 | tail -1
bash tools/profile_gpu.sh ns512u3 10 ${R}_ns512u3 | tail -1
ROMS_HIP_OVERLAP=0 bash tools/profile_gpu.sh ns512u3 10 ${R}_ns512u3_serial | tail -1
bash tools/profile_gpu.sh benchmark3 8 ${R}_b3 | tail -1
bash tools/profile_pmc.sh ns512u3 4 ${R}_ns512u3 | tail -2
bash tools/profile_pmc.sh benchmark1 6 ${R}_b1 | tail -2
