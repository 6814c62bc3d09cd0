#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call8; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
for w in mae clip16; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$w -o t -- $B --workload $w --steps 6 --warmup 4 > $O/prof_$w.log 2>&1
CSV=$(find /tmp/p_$w -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 4 adamw_kernel > $O/trace_timeline_$w.txt 2>&1
done
cd $GRAFT_REPO_ROOT
for w in mae clip16; do head -9 $O/trace_timeline_$w.txt; grep -A10 "idle before" $O/trace_timeline_$w.txt; tail -25 $O/trace_timeline_$w.txt; done
