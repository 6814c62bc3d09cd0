#!/bin/bash
# round 3, run 6: kernel breakdown of the ViT workloads with the 8-phase kernel in (where do MAE / CLIP-B/16 spend their step?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
cd /tmp
for w in mae clip16; do
  B="python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --no-kernel-timing"
  PASSL_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o s -- $B --steps 4 --warmup 2 > $O/prof_$w.log 2>&1
  python $T/rocpd_summary.py $(db /tmp/p_$w) 6 "PASSL_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 (6 steps in the trace)" > $O/kernel_stats_${w}_serial.txt 2>&1
  (cd $GRAFT_REPO_ROOT; timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench.jsonl 2>> $O/bench.err)
done
head -32 $O/kernel_stats_mae_serial.txt; head -32 $O/kernel_stats_clip16_serial.txt; cut -c1-300 $O/bench.jsonl
