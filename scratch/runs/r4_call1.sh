#!/bin/bash
# round 4, GPU call 1: (a) the prepared matrix-operand 8-phase A/B, (b) the ViT-side GPU tests with that form on,
# (c) driver-like baseline bench lines of this box (20 steps / 5 warm-up) incl. the key-pipeline switch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_call1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scratch/ab_8p_dense.py > $O/ab_8p_dense.log 2>&1; echo "exit $?" >> $O/ab_8p_dense.log
PASSL_IGEMM_8P_DENSE=2 timeout 900 python -m pytest tests/test_mae_gpu.py tests/test_clip_gpu.py tests/test_mocov3_gpu.py tests/test_layers_gpu.py tests/test_ops_gpu.py -m gpu -q -x > $O/tests_dense2.log 2>&1; echo "exit $?" >> $O/tests_dense2.log
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_20_$i.json 2> $O/bench_20_$i.err
done
PASSL_KEY_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_20_nokey.json 2> $O/bench_20_nokey.err
timeout 300 python bench.py --steps 50 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/bench_50.json 2> $O/bench_50.err
tail -3 $O/ab_8p_dense.log; tail -3 $O/tests_dense2.log; cat $O/bench_*.json | cut -c1-200
