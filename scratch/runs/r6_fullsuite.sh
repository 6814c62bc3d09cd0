#!/bin/bash
# the closing GPU suite exactly as the driver runs it, then smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_full; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
tail -8 $O/tests_gpu.log; tail -2 $O/smoke.log
