#!/bin/bash
# one-step timeline + kernel stats of the self-exchange run (mailbox): tools/gpu_debug/selfx_trace.sh [workload] [steps] [tag]
W=${1:-benchmark1}; N=${2:-20}; T=${3:-selfx}
R=$PWD; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/tr_$T
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/gpu_debug/gpu_selfx_prof.py $W $N peer > $O/log 2>&1
python $R/tools/trace_step.py $O/t_kernel_trace.csv > $O/step.txt; rm -f $O/t_kernel_trace.csv
grep selfx $O/log
cat $O/step.txt
