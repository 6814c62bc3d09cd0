#!/bin/bash
# stream priorities again, now that the side stream's launches are lighter (round 5 read them level)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c38; rm -rf $O; mkdir -p $O
run() { # label envs rep
  env $2 timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 rep $3: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2 3; do
  run default "A=1" $rep
  run main_high "PASSL_MAIN_PRIORITY=-1" $rep
  run main_default_stream0 "PASSL_MAIN_PRIORITY=0" $rep
  run halo3 "PASSL_OPTIONS=wgrad_halo_stages=3" $rep
done | tee $O/ab.txt
