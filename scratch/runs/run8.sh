#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run8
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "bn" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
timeout 900 python -m pytest tests/test_moco_gpu.py -q --tb=short -k "reproducible or small_fp32 or cfg1_bf16" > $O/moco.log 2>&1; echo "rc=$?" >> $O/moco.log
timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in ops moco; do tail -n 3 $O/$f.log; done; head -c 230 $O/bench.json
