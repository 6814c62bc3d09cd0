#!/bin/bash
# forced data-parallel path against the plain step, with the old (512) and the shipped weight-gradient grid targets, same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/c30; rm -rf $O; mkdir -p $O
run() { # label dpflag pipe halo rep
  PASSL_WGRAD_TARGET_BLOCKS=$3 PASSL_WGRAD_HALO_TARGET_BLOCKS=$4 timeout 300 python bench.py $2 --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 pipe $3 halo $4 rep $5: %.3f ms' % d['ms_per_step'], (d.get('dist') or {}).get('allreduce_exposed_ms'), (d.get('dist') or {}).get('collective_host_ms_per_step'))"
}
for rep in 1 2 3; do
  run plain "" 0 0 $rep
  run dp --dp-force 0 0 $rep
  run plain "" 512 512 $rep
  run dp --dp-force 512 512 $rep
done | tee $O/ab.txt
