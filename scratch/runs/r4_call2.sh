#!/bin/bash
# round 4, GPU call 2: the native step plan — unit tests, bit-identity tests, foreign-op survey per workload, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_call2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_ops_gpu.py::test_fill_copy_cast_unpad_kernels "tests/test_moco_gpu.py::test_step_plan_replay_is_bit_identical" tests/test_moco_gpu.py::test_trainer_replays_the_step_plan_by_default tests/test_moco_gpu.py::test_step_graph_replay_is_bit_identical tests/test_moco_gpu.py::test_step_is_bit_reproducible -m gpu -q -x -s > $O/tests_plan.log 2>&1; echo "exit $?" >> $O/tests_plan.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_plan.json 2> $O/bench_plan.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --eager > $O/bench_eager.json 2> $O/bench_eager.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_plan2.json 2> $O/bench_plan2.err
timeout 900 python scratch/plan_probe.py clip clip16 mae simclr --force > $O/probe.jsonl 2> $O/probe.err
timeout 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > $O/tests_dp.log 2>&1; echo "exit $?" >> $O/tests_dp.log
tail -5 $O/tests_plan.log; cut -c1-330 $O/bench_*.json; tail -3 $O/bench_plan.err; cut -c1-1500 $O/probe.jsonl; tail -5 $O/probe.err; tail -4 $O/tests_dp.log
