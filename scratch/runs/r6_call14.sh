#!/bin/bash
# round 6, call 14: the wave-per-patch 3x3 kernel: exactness, then time against the ring kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/c14; rm -rf $O; mkdir -p $O
K=tools/kbench
( echo "== check conv3x3_wave=1"; timeout 120 $K check conv3x3_wave=1 | grep -E "stage-1|8 x 8|CHECK|WRONG|wrong"
  echo "== check conv3x3_wave=0"; timeout 120 $K check conv3x3_wave=0 | grep -E "stage-1|8 x 8|CHECK"
  echo "== ab conv3x3_wave=0,1 (forward + statistics)"; timeout 120 $K ab conv3x3_wave=0,1 | head -4
  echo "== again"; timeout 120 $K ab conv3x3_wave=0,1 | head -3
) > $O/kbench.txt 2>&1
cat $O/kbench.txt
