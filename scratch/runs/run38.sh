#!/bin/bash
# final full GPU suite of the round (regenerates gpurun_out/parity_*.txt) + smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2_run38
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r2_run38/pytest_gpu.txt; cat gpurun_out/r2_run38/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r2_run38/smoke.txt
