#!/bin/bash
# round 4, GPU call 5: bias column sums + LayerNorm parameter fold on the side stream (ViT paths); what precedes the
# optimizer launch in a replayed step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_call5; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mae_gpu.py tests/test_mae_v2_gpu.py tests/test_clip_gpu.py tests/test_mocov3_gpu.py tests/test_layers_gpu.py tests/test_step_plan_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for w in mae clip16 clip; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_moco.json 2> $O/bench_moco.err
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv2 -o t -- $B --workload mae --steps 6 --warmup 4 > $O/prof_csv_mae.log 2>&1
CSV=$(ls /tmp/p_csv2/*/*kernel_trace.csv /tmp/p_csv2/*kernel_trace.csv 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 4 adamw_kernel > $O/trace_timeline_mae.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -4 $O/tests.log; cut -c1-230 $O/bench_workloads.jsonl; cut -c1-230 $O/bench_moco.json; tail -40 $O/trace_timeline.txt; head -12 $O/trace_timeline_mae.txt
