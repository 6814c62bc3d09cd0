#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c23; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for t in "0 0" "256 256" "256 192" "256 128" "512 256" "256 320"; do
  set -- $t
  PASSL_WGRAD_TARGET_BLOCKS=$1 PASSL_WGRAD_HALO_TARGET_BLOCKS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipe $1 halo $2 rep $rep: %.3f ms' % d['ms_per_step'])"
done; done | tee $O/ab.txt
