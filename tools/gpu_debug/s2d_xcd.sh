#!/bin/bash
# barotropic kernel with / without the XCD-aware block order (ROMS_HIP_S2D_XCD): per-launch time of the per-call kernel
# and the whole step of each workload.  Run ON THE GPU BOX.
export PYTHONPATH=$PWD
for x in 0 1; do
  for wl in ns512u3 benchmark3 benchmark2; do
    ROMS_HIP_S2D_XCD=$x python tools/gpu_debug/gpu_step2d_probe.py $wl 2>&1 | grep -E "^(pred|corr)" | sed "s/^/xcd=$x $wl /"
  done
  for wl in benchmark1 ns512u3 benchmark3; do
    ROMS_HIP_S2D_XCD=$x python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-north-star 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd=$x $wl ms_per_step', d['ms_per_step'])"
  done
done
