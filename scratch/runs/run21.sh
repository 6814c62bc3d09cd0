#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "pool" 2>&1 | tail -3
python scratch/bench_pool.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140
