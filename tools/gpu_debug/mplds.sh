#!/bin/bash
# ON THE GPU BOX: the fused MPDATA limiter (k_mp_lds.h) -- bits against the point-wise pair, config-5 step time and the
# kernels' serial durations, for a few chunk lengths
R=$PWD; export PYTHONPATH=$R
python -m pytest tests/test_gpu_parity.py -q -x -k "forms_agree and config5" 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -q -x -k "mpdata or config5_physics" 2>&1 | tail -2
for v in "0 10" "1 5" "1 10" "1 17" "1 50"; do set -- $v
  ROMS_HIP_MPLDS=$1 ROMS_HIP_MPLDS_KC=$2 python bench.py --workload config5 --steps 12 --warmup 3 --no-cpu-baseline --no-north-star --breakdown-file gpurun_out/mplds_bd.json 2>gpurun_out/mplds.err | grep '"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); b=json.load(open('gpurun_out/mplds_bd.json'))
print('MPLDS=$1 KC=$2 ms_per_step', round(d['ms_per_step'],4), {k:round(1e6*v['seconds']/v['launches'],1) for k,v in b.items() if 'mp_' in k})"
done
