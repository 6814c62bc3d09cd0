#!/bin/bash
cd $GRAFT_REPO_ROOT
export ONLY=64-64-1,64-256-1,256-64-1,256-128-1,128-512-1,256-512-1,256-1024-1
export PASSL_IGEMM_PERSIST_MIN_TILES=0
echo "== base"; python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
echo "== nontemporal stores"; PASSL_IGEMM_DBG=16 python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
echo "== BN=64 tiles"; PASSL_IGEMM_FORCE_BN64=1 python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
echo "== BN=64 tiles + nt"; PASSL_IGEMM_DBG=16 PASSL_IGEMM_FORCE_BN64=1 python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"
