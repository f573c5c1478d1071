R=$PWD; export PYTHONPATH=$R
B="python bench.py --workload benchmark1_mask --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown --no-north-star"
ms() { grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
for r in 1 2 3; do
  $B 2>&1 | ms "default"
  ROMS_HIP_LATE_MASK=1 $B 2>&1 | ms "LATE_MASK=1"
  ROMS_HIP_LOOP=0 $B 2>&1 | ms "LOOP=0"
done
