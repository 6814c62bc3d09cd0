#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call6; rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 $B --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
python $T/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
python $T/trace_chain.py $CSV 4 > $O/trace_chain.txt 2>&1
MAIN=$(grep -m1 "^stream" $O/trace_chain.txt | awk '{print $2}' | tr -d ':')
python $T/trace_chain.py $CSV 1 sgd_kernel --list $MAIN > $O/trace_chain_list.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 $B --steps 20 --warmup 5 > $O/bench2.json 2> $O/bench2.err
python - <<PY
import json
for f in ('bench1','bench2'):
    d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
head -36 $O/trace_chain.txt
