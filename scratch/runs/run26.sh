#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_moco_gpu.py -q -x -k "bn or reproducible or cfg1 or stat" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | cut -c1-150
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls /tmp/p_prof/*/*.db /tmp/p_prof/*.db 2>/dev/null | head -1) 10 x | head -40
