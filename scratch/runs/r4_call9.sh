#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call9; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_clip_gpu.py tests/test_step_plan_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for i in 1 2; do
timeout 400 python bench.py --workload clip16 --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 >> $O/bench.jsonl 2>> $O/bench.err
timeout 400 python bench.py --workload clip16 --eager --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 >> $O/bench.jsonl 2>> $O/bench.err
done
tail -4 $O/tests.log; python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r4_call9/bench.jsonl'):
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['step_launch'][:60])
PY
