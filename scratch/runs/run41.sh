#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mae_gpu.py -q -x 2>&1 | tail -2
python bench.py --workload mae --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 6 2>/dev/null | cut -c1-150
