#!/bin/bash
# full GPU suite + the other workloads' bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_run22
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
for w in simclr mae clip clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 600 python bench.py --workload clip16 --batch 1024 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
python - <<'PY'
import json
for l in open('gpurun_out/r2_run22/bench_workloads.jsonl'):
    d = json.loads(l)
    print(d['metric'][:60], d['value'], d['ms_per_step'], d['config'].get('hbm_reserved_gb'), d['config'].get('allocator_retries'), (d.get('roofline') or {}).get('frac'))
PY
