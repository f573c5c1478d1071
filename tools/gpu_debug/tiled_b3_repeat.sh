#!/bin/bash
# BENCHMARK3 in its 2x4 partition, eight processes sharing the device (tests/test_gpu_parity.py:
# test_baseline_configs_in_their_tiled_form_match_single_tile[benchmark3]), repeated; the ranks' own messages kept.
N=${1:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1 ROMS_HIP_PEER_TIMEOUT=30
SPEC='{"workload":"benchmark3","steps":3,"tiles":[2,4],"fields":["zeta","ubar"],"gpu":true,"probe":true,"transport":"peer"}'
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $((29900+i)) tests/mp/run_tiles.py /tmp/tb3_$i.npz "$SPEC" > gpurun_out/tb3_$i.log 2>&1
  echo "run $i rc=$? $(( $(date +%s) - t0 )) s: $(grep -v 'Gloo\|^W1\|^E1\|^ \|^$\|Traceback\|File \|====\|----\|time \|host \|rank \|exitcode\|error_file\|traceback\|torch.distributed\|Failures\|Root Cause\|run_tiles.py FAILED' gpurun_out/tb3_$i.log | head -4 | cut -c1-300 | tr '\n' '|')"
done
