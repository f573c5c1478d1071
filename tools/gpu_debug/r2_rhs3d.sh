#!/bin/bash
# A/B of the rhs3d_tile point part: LDS-tiled (min waves 2/3/4, chunk sizes) vs point-wise.  Run through gpurun.
OUT=gpurun_out/r2_rhs3d; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "column_kernel_forms or kernels_one_by_one or closed" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for v in "ROMS_HIP_RHS3D_LDS=0" "ROMS_HIP_RHS3D_W=2" "ROMS_HIP_RHS3D_W=3" "ROMS_HIP_RHS3D_W=4" "ROMS_HIP_RHS3D_W=3 ROMS_HIP_RHS3D_KC=50" "ROMS_HIP_RHS3D_W=3 ROMS_HIP_RHS3D_KC=13" "ROMS_HIP_RHS3D_W=2 ROMS_HIP_RHS3D_KC=50"; do
  echo "== $v"; env $v KB_ROWS=14 python tools/gpu_debug/gpu_kbreak.py ns512u3 4 2>&1 | grep "k_rhs3d_pt\|sum of\|async"
done
for v in "ROMS_HIP_RHS3D_LDS=0" "ROMS_HIP_RHS3D_W=3" "ROMS_HIP_RHS3D_W=2" "ROMS_HIP_RHS3D_W=3 ROMS_HIP_RHS3D_KC=10"; do
  echo "== b1 $v"; env $v KB_ROWS=30 python tools/gpu_debug/gpu_kbreak.py benchmark1 6 2>&1 | grep "k_rhs3d_pt\|sum of\|async"
done
