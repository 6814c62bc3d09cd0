#!/bin/bash
O=gpurun_out/r5_call3; mkdir -p $O
K=tools/kbench
{
  echo "== check"; timeout 60 $K check | tail -2
  echo "== fincheck"; timeout 90 $K fincheck
  echo "== fintime"; timeout 90 $K fintime
} > $O/kbench.txt 2>&1
tail -60 $O/kbench.txt
