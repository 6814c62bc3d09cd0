#!/bin/bash
# round 4, GPU call 4: new tests (v2 ViT / MAE front end, BN-MLP heads, plan with MAE), DP tests under the plan,
# stream-priority experiment on the bench, plan vs eager for clip / mae
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_call4; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_mae_v2_gpu.py tests/test_step_plan_gpu.py "tests/test_layers_gpu.py::test_bn_mlp_heads_teacher_forced_bf16" tests/test_layers_gpu.py::test_contrastive_criteria_fp32_kernels_against_float64 tests/test_dp_gpu.py -m gpu -q > $O/tests_new.log 2>&1; echo "exit $?" >> $O/tests_new.log
python - > $O/prio.txt 2>&1 <<'PY'
import torch
for p in (-2, -1, 0, 1, 2):
    print(p, '->', torch.cuda.Stream(priority=p).priority)
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'no priority_range')
PY
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-timing"
for i in 1 2; do
timeout 300 $B > $O/bench_base_$i.json 2> $O/bench_base_$i.err
PASSL_MAIN_PRIORITY=-1 timeout 300 $B > $O/bench_mainhi_$i.json 2> $O/bench_mainhi_$i.err
PASSL_AUX_PRIORITY=1 timeout 300 $B > $O/bench_auxlo_$i.json 2> $O/bench_auxlo_$i.err
done
PASSL_MAIN_PRIORITY=-1 PASSL_AUX_PRIORITY=1 timeout 300 $B > $O/bench_both.json 2> $O/bench_both.err
timeout 600 python scratch/plan_probe.py mae clip16 > $O/probe.jsonl 2> $O/probe.err
grep -v "^\s*$" $O/tests_new.log | tail -15; cat $O/prio.txt
for f in $O/bench_*.json; do echo $f; cut -c1-260 $f | sed 's/.*"value"/"value"/'; done; cut -c1-700 $O/probe.jsonl
