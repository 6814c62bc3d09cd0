#!/bin/bash
O=gpurun_out/r5_call3; mkdir -p $O
K=tools/kbench
{
  echo "== fincheck"; timeout 120 $K fincheck
  echo "== fintime"; timeout 90 $K fintime
  echo "== finstress"; timeout 200 $K finstress iters=45
} > $O/kbench.txt 2>&1
tail -60 $O/kbench.txt
