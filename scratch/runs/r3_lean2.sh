#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3g
S="64-64-1,64-256-1,256-64-1,256-128-1,128-512-1,256-1024-1"
for v in "PASSL_IGEMM_LEAN2=0" "PASSL_IGEMM_LEAN2=1" "PASSL_IGEMM_LEAN2_BN64=1"; do
  echo "== $v" >> gpurun_out/r3g/lean2.txt
  env $v ONLY=$S timeout 200 python scratch/bench_convs.py 2>&1 | grep -v "total\|amdgpu" >> gpurun_out/r3g/lean2.txt
done
cat gpurun_out/r3g/lean2.txt
