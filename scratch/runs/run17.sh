#!/bin/bash
cd $GRAFT_REPO_ROOT
export ONLY=64-64-1,64-256-1,256-64-1,256-128-1,128-512-1,256-512-1,256-1024-1
for d in 0 1 2 4 8 5; do echo "== PASSL_IGEMM_DBG=$d (1 no stores, 2 no epilogue, 4 no A loads, 8 no MFMA)"; PASSL_IGEMM_DBG=$d python scratch/bench_convs.py 2>&1 | grep -v "amdgpu.ids\|total"; done
