#!/bin/bash
# round 6, call 4 (second try): where do the 2.2 ms of the forced data-parallel path go?  kernel traces of both, chain
# view; also the R18 / MAE fine-tuning tests added since call 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/c4; rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
( timeout 600 python -m pytest tests/test_simclr_gpu.py tests/test_mae_gpu.py -q -x -k "r18 or finetune" 2>&1 | tail -15 ) > $O/tests_new.log 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_plain -o t -- $B --steps 8 --warmup 4 > $O/prof_plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_dp -o t -- $B --steps 8 --warmup 4 --dp-force > $O/prof_dp.log 2>&1
cd $GRAFT_REPO_ROOT
for w in plain dp; do
  CSV=$(ls /tmp/p_$w/*/*kernel_trace.csv /tmp/p_$w/*kernel_trace.csv 2>/dev/null | head -1)
  if [ -z "$CSV" ]; then echo "no trace for $w"; continue; fi
  timeout 120 python $T/trace_chain.py $CSV 4 > $O/trace_chain_$w.txt 2>&1
  timeout 120 python $T/trace_timeline.py $CSV 6 > $O/trace_timeline_$w.txt 2>&1
  MAIN=$(grep -m1 "^stream" $O/trace_chain_$w.txt | awk '{print $2}' | tr -d ':')
  timeout 120 python $T/trace_chain.py $CSV 1 sgd_kernel --list $MAIN > $O/trace_chain_list_$w.txt 2>&1
  timeout 60 gzip -c $CSV > $O/kernel_trace_$w.csv.gz
done
tail -5 $O/tests_new.log
head -30 $O/trace_chain_plain.txt; echo ======; head -45 $O/trace_chain_dp.txt
