#!/bin/bash
# ON THE GPU BOX: BENCHMARK1 ms/step under a few form switches (each twice)
R=$PWD; export PYTHONPATH=$R
run() { for i in 1 2; do env "$@" python bench.py --workload benchmark1 --steps 100 --warmup 10 --no-cpu-baseline --no-breakdown --no-north-star 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; done; }
run A=0
run ROMS_HIP_TADV_LDS=1
run ROMS_HIP_LMDCOL=1
run ROMS_HIP_LMDCOL=2
run ROMS_HIP_PRENEW_MARCH=1
run ROMS_HIP_EARLY_T3=0
run ROMS_HIP_UVCOL=1
run A=0
