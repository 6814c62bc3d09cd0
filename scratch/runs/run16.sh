#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/bench_bn.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_ops_gpu.py -q -x -k "bn" 2>&1 | tail -3
for u in 0 2 4 8; do PASSL_BN_STREAM_UNROLL=$u python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140; done
