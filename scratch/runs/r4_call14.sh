#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call14; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_clip_gpu.py tests/test_step_plan_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for rep in 1 2; do
for v in 1 0; do
for w in clip16 clip; do
  echo "PASSL_CLIP_TOWER_OVERLAP=$v $w" >> $O/ab.txt
  PASSL_CLIP_TOWER_OVERLAP=$v timeout 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 2>> $O/ab.err | cut -c1-200 >> $O/ab.txt
done; done; done
tail -4 $O/tests.log; cat $O/ab.txt | sed 's/"unit".*"ms_per_step"/ ms/' | sed 's/{"metric".*"value"/ value/' | cut -c1-90
