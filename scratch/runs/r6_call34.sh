#!/bin/bash
# eager steps read 0.1-0.3 ms FASTER than replayed ones in three evidence runs: is it WHEN the side stream gets its work?
cd $GRAFT_REPO_ROOT
O=gpurun_out/c34; rm -rf $O; mkdir -p $O
run() { # label envs flags rep
  env $2 timeout 300 python bench.py $3 --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 rep $4: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2 3; do
  run plan "A=1" "" $rep
  run eager "A=1" "--eager" $rep
  run plan_batch1 "PASSL_SIDE_BATCH=1" "" $rep
  run plan_batch8 "PASSL_SIDE_BATCH=8" "" $rep
  run plan_batch16 "PASSL_SIDE_BATCH=16" "" $rep
  run plan_batch64 "PASSL_SIDE_BATCH=64" "" $rep
  run eager_batch16 "PASSL_SIDE_BATCH=16" "--eager" $rep
done | tee $O/ab.txt
