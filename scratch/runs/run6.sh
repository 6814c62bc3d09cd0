#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run6
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "bn or conv_fwd" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
timeout 900 python -m pytest tests/test_moco_gpu.py -q --tb=short > $O/moco.log 2>&1; echo "rc=$?" >> $O/moco.log
timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 > $O/bench_overlap.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
PASSL_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 > $O/bench_nooverlap.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 > $O/bench_overlap2.json 2>> $O/bench.err
cp gpurun_out/parity_moco* $O/ 2>/dev/null
for f in ops moco; do tail -n 4 $O/$f.log; done; for f in overlap nooverlap overlap2; do head -c 230 $O/bench_$f.json; echo; done
