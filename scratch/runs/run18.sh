#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or dgrad" 2>&1 | tail -5
export ONLY=64-64-1,64-256-1,256-64-1,256-128-1,128-512-1,256-512-1,256-1024-1
for m in 0 2048 1024; do echo "== PASSL_IGEMM_PERSIST_MIN_TILES=$m"; PASSL_IGEMM_PERSIST_MIN_TILES=$m python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"; done
unset ONLY
for m in 0 2048 1024; do PASSL_IGEMM_PERSIST_MIN_TILES=$m python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140; done
