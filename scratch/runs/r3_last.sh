#!/bin/bash
# round 3, last GPU call: the SimSiam tests after the default-initialisation change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_last
rm -rf $O; mkdir -p $O
timeout 105 python -m pytest tests/test_simsiam_gpu.py "tests/test_linprobe_v2_gpu.py::test_pretrain_checkpoint_feeds_the_probe" -m gpu -q > $O/tests.log 2>&1
echo "exit $?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-250
