#!/bin/bash
# stem tail, second attempt: urgent hand-off of the first stage's weight gradients, last stem piece on the main stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call12; rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name', e); print(open('$O/$name.err').read()[-800:])
PY
}
OLD="PASSL_STEM_WGRAD_PARTS=1 PASSL_STEM_TAIL_FLUSH=0 PASSL_SIDE_URGENT_ROWS=0 PASSL_OPTIONS=stem_pool_form=0"
run old $OLD
run new A=1
run no_urgent PASSL_SIDE_URGENT_ROWS=0
run no_mainlast PASSL_STEM_WGRAD_MAIN_LAST=0
run parts1 PASSL_STEM_WGRAD_PARTS=1
run parts3 PASSL_STEM_WGRAD_PARTS=3
run urgent150k PASSL_SIDE_URGENT_ROWS=150000
run old2 $OLD
run new2 A=1
timeout 300 python -m pytest tests/test_moco_gpu.py tests/test_step_plan_gpu.py -m gpu -x -q -k "moco" > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -3 $O/tests.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_new -o t -- $B --steps 8 --warmup 4 > $O/prof_new.log 2>&1
CSV=$(ls /tmp/p_new/*/*kernel_trace.csv /tmp/p_new/*kernel_trace.csv 2>/dev/null | head -1)
python $T/trace_timeline.py $CSV 6 > $O/timeline_new.txt 2>&1
python $T/trace_chain.py $CSV 4 > $O/chain_new.txt 2>&1
head -3 $O/timeline_new.txt; tail -24 $O/timeline_new.txt
