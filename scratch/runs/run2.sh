#!/bin/bash
# round-2 run 2: validate the atomic-free statistics / wgrad / InfoNCE paths and the fused BN backward
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run2
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -x > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
timeout 900 python -m pytest tests/test_moco_gpu.py -q --tb=short > $O/moco.log 2>&1; echo "rc=$?" >> $O/moco.log
timeout 600 python -m pytest tests/test_simclr_gpu.py tests/test_clas_gpu.py -q --tb=short > $O/simclr.log 2>&1; echo "rc=$?" >> $O/simclr.log
timeout 600 python -m pytest tests/test_dp_gpu.py -q --tb=short -k "rccl or bench or moco" > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
cp gpurun_out/parity_* $O/ 2>/dev/null
tail -3 $O/ops.log $O/moco.log $O/simclr.log $O/dp.log; cat $O/bench.json | head -c 600
