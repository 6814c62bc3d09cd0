#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run12
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "2gb" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
for ov in 1 0; do
PASSL_OVERLAP=$ov timeout 400 python bench.py --workload clip16 --batch 1024 --steps 5 --warmup 6 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clip16 b1024 overlap=$ov', d['value'], d['ms_per_step'])" >> $O/ab.txt
PASSL_OVERLAP=$ov timeout 400 python bench.py --workload simclr --batch 512 --steps 5 --warmup 6 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('simclr b512 overlap=$ov', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
tail -n 4 $O/ops.log; cat $O/ab.txt
