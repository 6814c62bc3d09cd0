#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3_suite
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r3_suite/tests_gpu.log 2>&1
echo "exit $?" >> gpurun_out/r3_suite/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r3_suite/smoke.log 2>&1
tail -15 gpurun_out/r3_suite/tests_gpu.log; tail -2 gpurun_out/r3_suite/smoke.log
