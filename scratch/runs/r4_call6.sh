#!/bin/bash
# round 4, GPU call 6: same-box A/B of the side-stream reductions on the ViT workloads; what precedes the optimizer
# launch in a replayed step (kernel trace csv -> tools/trace_timeline.py)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_call6; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for rep in 1 2; do
for v in 1 0; do
for w in mae clip16; do
  echo "PASSL_SIDE_REDUCTIONS=$v $w" >> $O/ab.txt
  PASSL_SIDE_REDUCTIONS=$v timeout 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 2>> $O/ab.err | cut -c1-200 >> $O/ab.txt
done; done; done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(find /tmp/p_csv -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv2 -o t -- $B --workload mae --steps 6 --warmup 4 > $O/prof_csv_mae.log 2>&1
CSV=$(find /tmp/p_csv2 -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $CSV 4 adamw_kernel > $O/trace_timeline_mae.txt 2>&1
cd $GRAFT_REPO_ROOT
cat $O/ab.txt | sed 's/"unit".*"ms_per_step"/ ms/' | cut -c1-150; tail -42 $O/trace_timeline.txt; head -14 $O/trace_timeline_mae.txt; tail -26 $O/trace_timeline_mae.txt
