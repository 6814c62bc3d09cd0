#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call5; rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
python $T/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
python $T/trace_chain.py $CSV 4 > $O/trace_chain.txt 2>&1
MAIN=$(grep -m1 "^stream" $O/trace_chain.txt | awk '{print $2}' | tr -d ':')
python $T/trace_chain.py $CSV 1 sgd_kernel --list $MAIN > $O/trace_chain_list.txt 2>&1
head -50 $O/trace_chain.txt
