#!/bin/bash
cd $GRAFT_REPO_ROOT
export ONLY=64-64-1,64-256-1,256-64-1,256-128-1,128-512-1,256-512-1,256-1024-1
python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
unset ONLY
python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140
