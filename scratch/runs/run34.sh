#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mae_gpu.py tests/test_clip_gpu.py -q -x 2>&1 | tail -3
python bench.py --workload mae --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 6 2>/dev/null | cut -c1-150
python bench.py --workload clip --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 6 2>/dev/null | cut -c1-150
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_mae -o s -- python $GRAFT_REPO_ROOT/bench.py --workload mae --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls /tmp/p_mae/*/*.db /tmp/p_mae/*.db 2>/dev/null | head -1) 10 "rocprofv3 --kernel-trace --stats -- python bench.py --workload mae --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace)" > $GRAFT_REPO_ROOT/gpurun_out/r2_mae_kernel_stats.txt
head -24 $GRAFT_REPO_ROOT/gpurun_out/r2_mae_kernel_stats.txt
