#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
S="512-128-1,512-256-1,1024-256-1,256-1024-1,2048-512-1,512-2048-1,256-512-1,512-1024-1,1024-2048-1,1024-512-1,128-512-1"
for v in "igemm_8p=0" "igemm_8p=0,igemm_ring_bk=32,igemm_ring_stages32=3" "igemm_8p=0,igemm_ring_bk=32,igemm_ring_stages32=4" "igemm_8p=0,igemm_ring_min_nk=2" "igemm_8p=0,igemm_ring_min_nk=2,igemm_ring_bk=32,igemm_ring_stages32=3"; do
  echo "== $v" >> gpurun_out/r3i/ring32.txt
  PASSL_OPTS=$v ONLY=$S timeout 200 python scratch/bench_convs.py 2>&1 | grep -v "total\|amdgpu" | cut -c1-140 >> gpurun_out/r3i/ring32.txt
done
cat gpurun_out/r3i/ring32.txt
