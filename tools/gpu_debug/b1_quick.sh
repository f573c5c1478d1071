#!/bin/bash
# quick check ON THE GPU BOX: BENCHMARK1 ms/step (3 runs), bit-identity test, one-step timeline under rocprofv3
R=$PWD; export PYTHONPATH=$R
for i in 1 2 3; do python bench.py --workload benchmark1 --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown --no-north-star 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1 ms_per_step', d['ms_per_step'])"; done
python -m pytest tests/test_gpu_parity.py -q -x -k "bit_identical or benchmark1_full or benchmark_physics" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/tr_b1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_b1 -o t -- python3 $R/bench.py --workload benchmark1 --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown --no-north-star > $R/gpurun_out/tr_b1/log 2>&1
cd $R; python tools/trace_step.py gpurun_out/tr_b1/t_kernel_trace.csv > gpurun_out/tr_b1/step.txt; rm gpurun_out/tr_b1/t_kernel_trace.csv; head -40 gpurun_out/tr_b1/step.txt
