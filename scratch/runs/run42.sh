#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_run42
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_run42/bench_default_nocpu.json 2>/dev/null; cut -c1-200 gpurun_out/r2_run42/bench_default_nocpu.json
timeout 300 python bench.py --dtype fp32 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r2_run42/bench_fp32.json 2>/dev/null; cut -c1-200 gpurun_out/r2_run42/bench_fp32.json
