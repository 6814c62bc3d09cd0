#!/bin/bash
# round 3, call 2: full GPU suite with the 8-phase kernel on auto + workload benches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_run2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_moco.json 2> $O/bench_moco.err; cut -c1-200 $O/bench_moco.json
PASSL_IGEMM_8P=0 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-kernel-timing > $O/bench_moco_no8p.json 2>> $O/bench_moco.err; cut -c1-200 $O/bench_moco_no8p.json
for w in mae clip16 clip; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
  PASSL_IGEMM_8P=0 timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing >> $O/bench_workloads_no8p.jsonl 2>> $O/bench_workloads.err
done
cut -c1-160 $O/bench_workloads.jsonl; cut -c1-160 $O/bench_workloads_no8p.jsonl
