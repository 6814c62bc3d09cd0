#!/bin/bash
# round 3: kernel trace (csv, with stream ids) of the MoCo bench for a timeline / critical-path look
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_trace
rm -rf $O; mkdir -p $O
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -o t -- $B --steps 6 --warmup 2 > $O/prof.log 2>&1
ls -laR /tmp/p_tr | head -20
f=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
head -2 $f
gzip -c $f > $O/kernel_trace.csv.gz
ls -la $O
tail -2 $O/prof.log
