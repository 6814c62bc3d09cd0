#!/bin/bash
# weight-gradient grid rule as shipped (spatial layers 256, Linears 512) against 512 / 512, then the full GPU suite and smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/c27; rm -rf $O; mkdir -p $O
run() { # workload pipe halo rep
  PASSL_WGRAD_TARGET_BLOCKS=$2 PASSL_WGRAD_HALO_TARGET_BLOCKS=$3 timeout 300 python bench.py --workload $1 --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 pipe $2 halo $3 rep $4: %.3f ms' % d['ms_per_step'])"
}
for w in moco simclr mae clip16; do for rep in 1 2 3; do run $w 512 512 $rep; run $w 0 0 $rep; done; done | tee $O/ab.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
tail -8 $O/tests_gpu.log; tail -2 $O/smoke.log
