#!/bin/bash
# stem tail: second-form reduce pass, side-stream flush in front of the fused stem backward, piecewise stem weight gradient
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call10; rm -rf $O; mkdir -p $O
tools/kbench poolcheck > $O/poolcheck.txt 2>&1; echo "rc=$?" >> $O/poolcheck.txt; tail -2 $O/poolcheck.txt
tools/kbench pooltime > $O/pooltime.txt 2>&1; cat $O/pooltime.txt
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_moco_gpu.py tests/test_step_plan_gpu.py -m gpu -x -q -k "maxpool or moco or tail or stem" > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -5 $O/tests.log
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name', e); print(open('$O/$name.err').read()[-800:])
PY
}
run base PASSL_STEM_WGRAD_PARTS=1 PASSL_STEM_TAIL_FLUSH=0 PASSL_OPTIONS=stem_pool_form=0
run flush PASSL_STEM_WGRAD_PARTS=1 PASSL_OPTIONS=stem_pool_form=0
run flush_form1 PASSL_STEM_WGRAD_PARTS=1
run all A=1
run parts4 PASSL_STEM_WGRAD_PARTS=4
run base2 PASSL_STEM_WGRAD_PARTS=1 PASSL_STEM_TAIL_FLUSH=0 PASSL_OPTIONS=stem_pool_form=0
run all2 A=1
