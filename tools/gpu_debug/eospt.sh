cd /tmp && export TMPDIR=/tmp
for r in 0 1; do
  export ROMS_HIP_EOSPT=$r
  for ov in 0 1; do
    export ROMS_HIP_OVERLAP=$ov
    rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/eospt_${r}_$ov -o t -- python3 /root/repo/bench.py --workload benchmark1 --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-breakdown > /root/repo/gpurun_out/eospt_${r}_$ov.log 2>&1
    echo "EOSPT=$r OVERLAP=$ov: $(grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/eospt_${r}_$ov.log)"
    grep "k_eos" /root/repo/gpurun_out/eospt_${r}_$ov/t_kernel_stats.csv | cut -d, -f1-4
  done
done
unset ROMS_HIP_OVERLAP
for r in 0 1 0 1; do ROMS_HIP_EOSPT=$r python3 /root/repo/bench.py --workload benchmark1 --steps 100 --warmup 20 --no-cpu-baseline --no-north-star --no-breakdown 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/plain EOSPT=$r /"; done
