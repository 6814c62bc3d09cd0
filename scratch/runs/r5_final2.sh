#!/bin/bash
# last call of round 5: the full GPU suite and the smoke test on the final tree, then the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_final2
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
tail -4 $O/tests_gpu.log; tail -2 $O/smoke.log; head -c 400 $O/bench_moco.json; echo
