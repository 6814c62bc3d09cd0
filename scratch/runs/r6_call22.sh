#!/bin/bash
# fewer weight-gradient slices = less slab traffic (3.4 GB of the step's 100): same-box A/B of the grid target
cd $GRAFT_REPO_ROOT
O=gpurun_out/c22; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for t in 0 128 192 256 320; do
  PASSL_WGRAD_TARGET_BLOCKS=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('target $t rep $rep: %.3f ms' % d['ms_per_step'])"
done; done | tee $O/ab.txt
