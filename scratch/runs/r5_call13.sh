#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call13; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "dgrad_fused" > $O/tests1.log 2>&1; echo "exit $?" >> $O/tests1.log; tail -4 $O/tests1.log
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run new A=1
run off PASSL_FUSED_BN_BACKWARD2=0
run new2 A=1
run off2 PASSL_FUSED_BN_BACKWARD2=0
timeout 900 python -m pytest tests/test_moco_gpu.py tests/test_layers_gpu.py tests/test_step_plan_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; echo "exit $?" >> $O/tests2.log; tail -4 $O/tests2.log
