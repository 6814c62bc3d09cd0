#!/bin/bash
# in-kernel BatchNorm finalize: bit-exact conv outputs, tail vs separate launches, timings, stress under load
O=gpurun_out/r5_call2; mkdir -p $O
K=tools/kbench
{
  echo "== check (no tail)"; timeout 60 $K check | tail -3
  echo "== check tail=1"; timeout 90 $K check tail=1
  echo "== tailtime"; timeout 60 $K tailtime
  echo "== tailstress"; timeout 200 $K tailstress iters=60
} > $O/kbench.txt 2>&1
tail -80 $O/kbench.txt
