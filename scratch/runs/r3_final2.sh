#!/bin/bash
# round 3, closing run: the full GPU suite of the final code + smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final2
rm -rf $O; mkdir -p $O
timeout 560 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1
echo "exit $?" >> $O/tests_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
tail -8 $O/tests_gpu.log | cut -c1-250; tail -2 $O/smoke.log
