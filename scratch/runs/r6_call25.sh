#!/bin/bash
# the intensity-based weight-gradient grid rule (default) against the fixed 512 of rounds 1-5, every workload, same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; rm -rf $O; mkdir -p $O
for w in moco mae clip16 clip simclr linprobe; do for rep in 1 2 3; do for t in "512 512" "0 0"; do
  set -- $t
  PASSL_WGRAD_TARGET_BLOCKS=$1 PASSL_WGRAD_HALO_TARGET_BLOCKS=$2 timeout 300 python bench.py --workload $w --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w pipe $1 halo $2 rep $rep: %.3f ms' % d['ms_per_step'])"
done; done; done | tee $O/ab.txt
