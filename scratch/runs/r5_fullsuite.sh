#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_fullsuite; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
tail -15 $O/tests_gpu.log; tail -3 $O/smoke.log
