#!/bin/bash
# last call of round 5 (second edition): the full GPU suite on the final tree, then smoke and the bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_final3
rm -rf $O; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
tail -4 $O/tests_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
head -c 600 $O/bench_moco.json; echo
