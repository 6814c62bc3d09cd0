#!/bin/bash
# the plan's pool is released after recording: reset test, SimCLR at 512 / GPU, the step-plan suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/c20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_step_plan_gpu.py -q -m gpu -s > $O/step_plan.log 2>&1; grep -n "reserved before\|passed\|failed\|Error" $O/step_plan.log | head
timeout 900 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_simclr_bs512.json 2> $O/bench_simclr_bs512.err
tail -3 $O/bench_simclr_bs512.err | cut -c1-400; grep '^{' $O/bench_simclr_bs512.json | cut -c1-400
