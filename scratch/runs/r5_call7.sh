#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call7; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "bn_relu_maxpool or maxpool or bn_fwd_bwd" > $O/tests1.log 2>&1; echo "exit $?" >> $O/tests1.log
tail -4 $O/tests1.log
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
timeout 300 $B > $O/bench_new.json 2> $O/bench_new.err
PASSL_FUSED_STEM_POOL=0 timeout 300 $B > $O/bench_nofuse.json 2> $O/bench_nofuse.err
timeout 300 $B > $O/bench_new2.json 2>> $O/bench_new.err
timeout 900 python -m pytest tests/test_moco_gpu.py tests/test_step_plan_gpu.py tests/test_simclr_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; echo "exit $?" >> $O/tests2.log
for f in bench_new bench_nofuse bench_new2; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])
except Exception as e: print('$f', e); print(open('$O/$f.err').read()[-1500:])
PY
done
tail -5 $O/tests2.log
