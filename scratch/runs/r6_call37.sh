#!/bin/bash
# moving inputs: what costs the 2 % — the host -> device transfer itself or the copy into the plan's input?
cd $GRAFT_REPO_ROOT
O=gpurun_out/c37; rm -rf $O; mkdir -p $O
run() { # label envs rep
  env $2 timeout 300 python bench.py --fresh-batches 3 --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 rep $3: resident %.3f ms  moving %.3f ms  ratio %.4f' % (d['ms_per_step'], d['fresh_inputs']['ms_per_step'], d['fresh_inputs']['ratio_to_resident']))"
}
for rep in 1 2 3; do
  run default "A=1" $rep
  run no_h2d "PASSL_RING_DIAG=noh2d" $rep
  run queues5 "GPU_MAX_HW_QUEUES=5" $rep
done | tee $O/ab.txt
