#!/bin/bash
# round 4, GPU call 3: native step plan after the foreign launches were removed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_call3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest "tests/test_moco_gpu.py::test_step_plan_replay_is_bit_identical" tests/test_step_plan_gpu.py -m gpu -q -s > $O/tests_plan.log 2>&1; echo "exit $?" >> $O/tests_plan.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_plan$i.json 2> $O/bench_plan$i.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --eager > $O/bench_eager$i.json 2> $O/bench_eager$i.err
done
timeout 900 python scratch/plan_probe.py moco clip mae > $O/probe.jsonl 2> $O/probe.err
grep -v "^\s*$" $O/tests_plan.log | tail -12; cut -c1-330 $O/bench_*.json; cut -c1-900 $O/probe.jsonl; tail -5 $O/probe.err
