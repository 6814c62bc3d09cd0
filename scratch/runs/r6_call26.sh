#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c26; rm -rf $O; mkdir -p $O
run() { # workload pipe halo intensity rep
  PASSL_WGRAD_TARGET_BLOCKS=$2 PASSL_WGRAD_HALO_TARGET_BLOCKS=$3 PASSL_WGRAD_INTENSITY=$4 timeout 300 python bench.py --workload $1 --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 pipe $2 halo $3 intensity $4 rep $5: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2 3 4; do
  run moco 512 512 250 $rep; run moco 0 0 250 $rep; run moco 256 256 250 $rep; run moco 0 0 450 $rep; run moco 0 0 150 $rep
done | tee $O/ab.txt
for rep in 1 2 3; do
  run mae 0 0 250 $rep; run mae 0 0 450 $rep; run clip16 0 0 250 $rep; run clip16 0 0 450 $rep
done | tee -a $O/ab.txt
