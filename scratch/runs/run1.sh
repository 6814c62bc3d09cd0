#!/bin/bash
# round-2 run 1: new tests (RCCL world 1, bench launch forms), default bench, kernel stats, MFMA-busy PMC pass
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run1
mkdir -p $O
timeout 600 python -m pytest tests/test_dp_gpu.py -x -q -k "rccl or bench" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
rocprofv3 -L > $O/counters.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $O $O/prof $O/pmc_mfma >> $O/tests.log 2>&1
tail -3 $O/tests.log; cat $O/bench.json | head -c 1500
