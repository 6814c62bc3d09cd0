#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | cut -c1-150
