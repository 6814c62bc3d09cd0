#!/bin/bash
# collectives issued from the side stream (default now): DP tests, then forced-DP against plain with variants, same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/c33; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu > $O/tests_dp.log 2>&1; tail -2 $O/tests_dp.log
run() { # label envs flags rep
  env $2 timeout 300 python bench.py $3 --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); x=d.get('dist') or {}; print('moco $1 rep $4: %.3f ms' % d['ms_per_step'], x.get('allreduce_exposed_ms'), x.get('collective_host_ms_per_step'))"
}
for rep in 1 2 3; do
  run plain "A=1" "" $rep
  run dp_side "A=1" "--dp-force" $rep
  run dp_own "PASSL_DP_COMM_STREAM=own" "--dp-force" $rep
  run dp_side_thread "PASSL_DP_THREAD=1" "--dp-force" $rep
  run dp_side_4buckets "PASSL_DP_BUCKETS=4" "--dp-force" $rep
  run dp_side_1bucket "PASSL_DP_BUCKETS=1" "--dp-force" $rep
done | tee $O/ab.txt
