#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c24; rm -rf $O; mkdir -p $O
for rep in 1 2; do for t in "0 0" "256 256" "192 256" "320 256" "384 256"; do
  set -- $t
  PASSL_WGRAD_TARGET_BLOCKS=$1 PASSL_WGRAD_HALO_TARGET_BLOCKS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco pipe $1 halo $2 rep $rep: %.3f ms' % d['ms_per_step'])"
done; done | tee $O/ab.txt
for w in mae clip16 simclr linprobe; do for rep in 1 2; do for t in "0 0" "256 256" "384 256"; do
  set -- $t
  PASSL_WGRAD_TARGET_BLOCKS=$1 PASSL_WGRAD_HALO_TARGET_BLOCKS=$2 timeout 300 python bench.py --workload $w --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w pipe $1 halo $2 rep $rep: %.3f ms' % d['ms_per_step'])"
done; done; done | tee -a $O/ab.txt
