cd /tmp && export TMPDIR=/tmp
for w in benchmark1 ns512u3 benchmark3; do
  for r in 0 1; do
    export ROMS_HIP_UVREG=$r
    export ROMS_HIP_OVERLAP=0
    rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/uvreg_${w}_$r -o t -- python3 /root/repo/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-north-star --no-breakdown > /root/repo/gpurun_out/uvreg_${w}_$r.log 2>&1
    echo "$w UVREG=$r: $(grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/uvreg_${w}_$r.log) $(grep 'k_s3uv_col' /root/repo/gpurun_out/uvreg_${w}_$r/t_kernel_stats.csv | cut -d, -f1-4)"
  done
done
