#!/bin/bash
# round 4, GPU call 10: side-stream hand-offs in batches (PASSL_SIDE_BATCH), bn backward-reduce grid, grouped reducers
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call10; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_moco_gpu.py tests/test_step_plan_gpu.py tests/test_dp_gpu.py tests/test_mae_gpu.py tests/test_layers_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for rep in 1 2; do
for v in 1 4 8; do
for w in moco mae; do
  echo "PASSL_SIDE_BATCH=$v $w" >> $O/ab.txt
  PASSL_SIDE_BATCH=$v timeout 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 2>> $O/ab.err | cut -c1-200 >> $O/ab.txt
done; done; done
for v in 1 4; do
  echo "PASSL_SIDE_BATCH=$v clip16" >> $O/ab.txt
  PASSL_SIDE_BATCH=$v timeout 400 python bench.py --workload clip16 --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 2>> $O/ab.err | cut -c1-200 >> $O/ab.txt
done
tail -4 $O/tests.log; cat $O/ab.txt | sed 's/"unit".*"ms_per_step"/ ms/' | sed 's/{"metric".*"value"/ value/' | cut -c1-90
