#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c15; rm -rf $O; mkdir -p $O
K=tools/kbench
( echo "== check"; timeout 120 $K check conv3x3_wave=1 | grep -E "stage-1|8 x 8|CHECK|WRONG|wrong"
  for d in 1 2 4 8 6 10 12 14; do echo "== dbg=$d"; timeout 60 $K ab conv3x3_wave_dbg=0,$d | sed -n 3p; done
  echo "== vs ring"; timeout 60 $K ab conv3x3_wave=0,1 | sed -n 3p
) > $O/kbench.txt 2>&1
cat $O/kbench.txt
