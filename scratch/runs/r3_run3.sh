#!/bin/bash
# round 3, run 3: validate graph / device-hyper / dist work; bench eager vs graph
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -x -q -k "step_graph or bench_two_ranks or test_dp_gpu or test_optim or engine_train or reproducible" > gpurun_out/r3c/tests_sel.log 2>&1
echo "sel exit $?" >> gpurun_out/r3c/tests_sel.log
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r3c/bench_eager.json 2> gpurun_out/r3c/bench_eager.err
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --graph > gpurun_out/r3c/bench_graph.json 2> gpurun_out/r3c/bench_graph.err
tail -3 gpurun_out/r3c/tests_sel.log; cat gpurun_out/r3c/bench_eager.json gpurun_out/r3c/bench_graph.json
