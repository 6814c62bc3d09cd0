#!/bin/bash
cd /tmp
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_run40; mkdir -p $O
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
for w in mae; do
  PASSL_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o s -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(db /tmp/p_$w) 10 "PASSL_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace; no side stream: every kernel's duration is its own)" > $O/${w}_kernel_stats_serial.txt
done
head -16 $O/mae_kernel_stats_serial.txt
