#!/bin/bash
# the 2x2-process part of tests/test_gpu_parity.py:test_pair_launches_hand_their_rim_across_tile_edges, repeated: how often does a
# rank give up waiting when four processes share the device?  usage: pair_rim_shared_repeat.sh <repeats> <probe 0|1> [env ...]
N=${1:-6}; PROBE=${2:-1}; shift 2
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1 ROMS_HIP_PEER_TIMEOUT=10 ROMS_HIP_LOOP=0 ROMS_HIP_PAIR_RIM=1 ROMS_HIP_LOOP_TIMEOUT=2
for kv in "$@"; do export "$kv"; done
P=$([ "$PROBE" = 1 ] && echo true || echo false)
SPEC='{"workload":"benchmark1","steps":4,"tiles":[2,2],"fields":["zeta","ubar"],"gpu":true,"probe":'$P',"transport":"peer"}'
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29800+i)) tests/mp/run_tiles.py /tmp/prs_$i.npz "$SPEC" > /tmp/prs_$i.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "run $i probe=$PROBE $* rc=$rc $(python3 -c "print(round($t1-$t0,1))") s $(grep -o 'exit_flag=.\{0,140\}' /tmp/prs_$i.log | head -2 | tr '\n' '|')"
done
