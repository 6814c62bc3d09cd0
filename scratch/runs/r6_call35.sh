#!/bin/bash
# transformer workloads: more than two weight-gradient workgroups per CU?
cd $GRAFT_REPO_ROOT
O=gpurun_out/c35; rm -rf $O; mkdir -p $O
run() { # workload pipe rep
  PASSL_WGRAD_TARGET_BLOCKS=$2 timeout 300 python bench.py --workload $1 --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 linear-target $2 rep $3: %.3f ms' % d['ms_per_step'])"
}
for w in mae clip16; do for rep in 1 2; do for t in 0 640 768 1024; do run $w $t $rep; done; done; done | tee $O/ab.txt
