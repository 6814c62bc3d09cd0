#!/bin/bash
# round 6, call 1: new tests + DP-forced / fresh-batches / fp32 bench lines on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_baseline_shapes_gpu.py tests/test_simclr_gpu.py -q -x -k "baseline or r18 or ntxent or clip_cross or mae_masking or attention_fwd_bwd_at" 2>&1 | tail -40 ) > gpurun_out/c1/tests_new.log 2>&1
( timeout 900 python -m pytest tests/test_dp_gpu.py -q -x 2>&1 | tail -30 ) > gpurun_out/c1/tests_dp.log 2>&1
( timeout 600 python -m pytest tests/test_step_plan_gpu.py tests/test_moco_gpu.py -q -x 2>&1 | tail -15 ) > gpurun_out/c1/tests_plan.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fresh-batches 3 > gpurun_out/c1/bench_plain.json 2> gpurun_out/c1/bench_plain.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-force > gpurun_out/c1/bench_dp_forced.json 2> gpurun_out/c1/bench_dp_forced.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > gpurun_out/c1/bench_plain2.json 2> gpurun_out/c1/bench_plain2.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-force --roofline-steps 0 > gpurun_out/c1/bench_dp_forced2.json 2> gpurun_out/c1/bench_dp_forced2.err
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype fp32 > gpurun_out/c1/bench_fp32.json 2> gpurun_out/c1/bench_fp32.err
tail -3 gpurun_out/c1/tests_new.log gpurun_out/c1/tests_dp.log gpurun_out/c1/tests_plan.log
for f in plain dp_forced plain2 dp_forced2 fp32; do python - <<PY
import json
try:
    z=json.loads(open('gpurun_out/c1/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', z['value'], z['ms_per_step'], z.get('value_fresh_inputs'), (z.get('dist') or {}).get('allreduce_exposed_ms'))
except Exception as e:
    print('$f FAILED', e)
PY
done
