#!/bin/bash
# timeline of the step's tail with the round's stem-tail changes on (defaults) and off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call11; rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
tr() { name=$1; shift; env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$name -o t -- $B --steps 8 --warmup 4 > $O/prof_$name.log 2>&1
  CSV=$(ls /tmp/p_$name/*/*kernel_trace.csv /tmp/p_$name/*kernel_trace.csv 2>/dev/null | head -1)
  python $T/trace_timeline.py $CSV 6 > $O/timeline_$name.txt 2>&1
  python $T/trace_chain.py $CSV 4 > $O/chain_$name.txt 2>&1
}
tr new A=1
tr old PASSL_STEM_WGRAD_PARTS=1 PASSL_STEM_TAIL_FLUSH=0 PASSL_OPTIONS=stem_pool_form=0
tail -32 $O/timeline_new.txt
