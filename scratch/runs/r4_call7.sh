#!/bin/bash
# round 4, GPU call 7: one end-of-backward join per backward pass (was one per hand-off)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call7; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_moco_gpu.py tests/test_step_plan_gpu.py tests/test_dp_gpu.py tests/test_simclr_gpu.py tests/test_mae_gpu.py tests/test_simsiam_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing >> $O/bench_moco.jsonl 2>> $O/bench.err
done
for w in mae clip16 clip simclr; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench.err
done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(find /tmp/p_csv -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -4 $O/tests.log; cut -c1-230 $O/bench_moco.jsonl; cut -c1-230 $O/bench_workloads.jsonl; sed -n 1,9p $O/trace_timeline.txt; grep -A8 "idle before" $O/trace_timeline.txt
