#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run9
mkdir -p $O
for ov in 0 1; do for fa in 0 1; do
  for rep in 1 2; do
  PASSL_OVERLAP=$ov PASSL_BN_FAST=$fa timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap=$ov fast=$fa rep=$rep', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done; done
cat $O/ab.txt
