cd /tmp && export TMPDIR=/tmp
export ROMS_HIP_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/tl_$1 -o t -- python3 /root/repo/bench.py --workload ns512u3 --steps 10 --warmup 3 --no-cpu-baseline --no-north-star --no-breakdown > /root/repo/gpurun_out/tl_$1.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/tl_$1.log
grep "k_tadv_lds\|k_rhs3d_lds" /root/repo/gpurun_out/tl_$1/t_kernel_stats.csv | cut -c1-100
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "column_kernel_forms" 2>&1 | tail -2
