#!/bin/bash
# which stream issues the gradient collectives (hardware-queue sharing with the product streams?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c32; rm -rf $O; mkdir -p $O
run() { # label envs rep
  env $2 timeout 300 python bench.py --dp-force --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco dp $1 rep $3: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco plain rep $rep: %.3f ms' % d['ms_per_step'])"
  run default "A=1" $rep
  run on_side "PASSL_DP_COMM_ON_SIDE=1" $rep
  run skip1 "PASSL_DP_COMM_SKIP=1" $rep
  run skip2 "PASSL_DP_COMM_SKIP=2" $rep
  run skip3 "PASSL_DP_COMM_SKIP=3" $rep
  run thread "PASSL_DP_THREAD=1" $rep
  run eager_comm "PASSL_DIST_EAGER=1" $rep
done | tee $O/ab.txt
