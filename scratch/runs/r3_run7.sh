#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
timeout 200 python scratch/bench_attn.py 2>&1 | grep -v amdgpu > gpurun_out/r3h/attn.txt
timeout 600 python -m pytest tests/test_mae_gpu.py tests/test_clip_gpu.py tests/test_layers_gpu.py -m gpu -q -k "attention or vit_block or reproducible or golden" > gpurun_out/r3h/tests.log 2>&1
for w in mae clip16 clip; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> gpurun_out/r3h/bench.jsonl 2>> gpurun_out/r3h/bench.err
done
cat gpurun_out/r3h/attn.txt; tail -3 gpurun_out/r3h/tests.log; cut -c1-260 gpurun_out/r3h/bench.jsonl
