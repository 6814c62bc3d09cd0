#!/bin/bash
# round 3: v2 linear-probe row (tests + parity reports)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_lp
rm -rf $O; mkdir -p $O
timeout 420 python -m pytest tests/test_linprobe_v2_gpu.py tests/test_abi.py "tests/test_mocov3_gpu.py::test_v2_engine_trains_mocov3_from_yaml" "tests/test_dp_gpu.py::test_two_ranks_linear_probe_engine_equals_one_rank_on_the_joint_batches" -m gpu -q -x > $O/tests.log 2>&1
echo "exit $?" >> $O/tests.log
cp gpurun_out/parity_lp_* $O/ 2>/dev/null
tail -40 $O/tests.log | cut -c1-300
