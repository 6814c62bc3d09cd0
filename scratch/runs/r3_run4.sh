#!/bin/bash
# round 3, run 4: MoCo-v3 parity tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_mocov3_gpu.py tests/test_dp_gpu.py -m gpu -q -k "mocov3" > gpurun_out/r3d/tests_mocov3.log 2>&1
echo "exit $?" >> gpurun_out/r3d/tests_mocov3.log
tail -60 gpurun_out/r3d/tests_mocov3.log
