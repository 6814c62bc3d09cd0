#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run11
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "2gb or conv_fwd or ring" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
timeout 600 python -m pytest tests/test_clas_gpu.py -q --tb=short -k "frozen" > $O/clas.log 2>&1; echo "rc=$?" >> $O/clas.log
timeout 400 python bench.py --workload simclr --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/simclr512.json 2> $O/simclr512.err; echo "rc=$?" >> $O/simclr512.err
timeout 400 python bench.py --workload clip16 --batch 1024 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/clip16_1024.json 2> $O/clip16_1024.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_clip -o s -- python $GRAFT_REPO_ROOT/bench.py --workload clip16 --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/prof_clip.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 4 $O/ops.log $O/clas.log; head -c 250 $O/simclr512.json; echo; head -c 250 $O/clip16_1024.json
