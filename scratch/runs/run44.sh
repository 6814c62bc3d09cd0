#!/bin/bash
# hypothesis test: dense single-K-tile igemm at 4 workgroups per CU (LEAN variant, 116 VGPRs, no spills)
cd $GRAFT_REPO_ROOT
export ONLY=64-64-1,64-256-1
echo "== 3 per CU"; python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
echo "== LEAN 4 per CU"; PASSL_IGEMM_LEAN=1 python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
unset ONLY
PASSL_IGEMM_LEAN=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fwd_dgrad_wgrad or stem_kernel or bn_fwd" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140
PASSL_IGEMM_LEAN=1 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 6 2>/dev/null | cut -c1-140
