#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c16; rm -rf $O; mkdir -p $O
K=tools/kbench
( for r in 8 4; do
    echo "== rows=$r check"; timeout 120 $K check conv3x3_wave=1 conv3x3_wave_rows=$r | grep -E "stage-1|8 x 8|CHECK|WRONG|wrong"
    echo "== rows=$r vs ring"; timeout 60 $K ab conv3x3_wave=0,1 conv3x3_wave_rows=$r | sed -n 3p
    for d in 2 4 6 14; do echo "== rows=$r dbg=$d"; timeout 60 $K ab conv3x3_wave_dbg=0,$d conv3x3_wave_rows=$r | sed -n 3p; done
  done
  echo "== vtime-like: relu / stats / bnb epilogues, rows 4 then ring"; 
) > $O/kbench.txt 2>&1
cat $O/kbench.txt
