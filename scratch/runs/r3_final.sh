#!/bin/bash
# round 3, final: bench line + kernel trace of the final code, then the full GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
timeout 600 python bench.py > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 2 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 10 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace; key pipeline and side stream on: durations of co-running kernels overlap)" > $O/kernel_stats.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1
echo "exit $?" >> $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
head -c 600 $O/bench_moco.json; echo; tail -6 $O/tests_gpu.log; tail -2 $O/smoke.log
