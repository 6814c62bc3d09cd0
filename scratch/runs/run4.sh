#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run4
mkdir -p $O
timeout 300 python scratch/dbg_stem_bias.py > $O/stem.log 2>&1
timeout 600 python -m pytest tests/test_moco_gpu.py -q --tb=short -k "v2_train or cfg1_bf16" > $O/moco.log 2>&1; echo "rc=$?" >> $O/moco.log
timeout 600 python -m pytest tests/test_dp_gpu.py -q --tb=short -k "rccl" > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
tail -n 12 $O/stem.log; tail -n 5 $O/moco.log $O/dp.log
