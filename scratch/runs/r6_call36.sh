#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c36; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu -k "dedicated or side_stream or rccl" > $O/tests_dp.log 2>&1; tail -3 $O/tests_dp.log
