#!/bin/bash
# round 6, call 2: persistent register-staged kernel (check + A/B), R18 goldens, DP-forced bench debug
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
K=tools/kbench
( echo "== check igemm_persist=0"; $K check igemm_persist=0 | tail -25
  echo "== check igemm_persist=1 igemm_persist_grid=8"; $K check igemm_persist=1 igemm_persist_grid=8 | tail -25
  echo "== check igemm_persist=1 (auto grid)"; $K check igemm_persist=1 | tail -3
  echo "== ab igemm_persist=0,1"; $K ab igemm_persist=0,1
  echo "== ab again"; $K ab igemm_persist=0,1 | tail -8
  echo "== vtime persist=0"; $K vtime igemm_persist=0
  echo "== vtime persist=1"; $K vtime igemm_persist=1
  echo "== ablate persist=1"; $K ablate igemm_persist=1
  echo "== ablate persist=0"; $K ablate igemm_persist=0
) > gpurun_out/c2/kbench.txt 2>&1
( timeout 600 python -m pytest tests/test_simclr_gpu.py -q -x -k "r18" 2>&1 | tail -15 ) > gpurun_out/c2/tests_r18.log 2>&1
timeout 300 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-force > gpurun_out/c2/bench_dp_forced.json 2> gpurun_out/c2/bench_dp_forced.err; echo "dp_forced rc=$?" > gpurun_out/c2/rc.txt
PASSL_PLAN=0 timeout 300 python -X faulthandler bench.py --steps 10 --warmup 3 --no-cpu-baseline --dp-force --roofline-steps 0 > gpurun_out/c2/bench_dp_forced_eager.json 2> gpurun_out/c2/bench_dp_forced_eager.err; echo "dp_forced_eager rc=$?" >> gpurun_out/c2/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > gpurun_out/c2/bench_persist1.json 2> gpurun_out/c2/bench_persist1.err
PASSL_IGEMM_PERSIST=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > gpurun_out/c2/bench_persist0.json 2> gpurun_out/c2/bench_persist0.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > gpurun_out/c2/bench_persist1b.json 2> gpurun_out/c2/bench_persist1b.err
PASSL_IGEMM_PERSIST=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > gpurun_out/c2/bench_persist0b.json 2> gpurun_out/c2/bench_persist0b.err
cat gpurun_out/c2/rc.txt
tail -4 gpurun_out/c2/tests_r18.log
grep -E "CHECK|WRONG|NOT RUN" gpurun_out/c2/kbench.txt | head
grep -A16 "== ab igemm_persist" gpurun_out/c2/kbench.txt | head -40
for f in persist1 persist0 persist1b persist0b dp_forced dp_forced_eager; do python - <<PY
import json
try:
    z=json.loads(open('gpurun_out/c2/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', z['value'], z['ms_per_step'], (z.get('dist') or {}).get('allreduce_exposed_ms'))
except Exception as e:
    print('$f FAILED', e)
PY
done
