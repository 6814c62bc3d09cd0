#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run10
mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --tb=short > $O/all.log 2>&1; echo "rc=$?" >> $O/all.log
timeout 400 python bench.py --workload simclr --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/simclr512.json 2> $O/simclr512.err; echo "rc=$?" >> $O/simclr512.err
timeout 400 python bench.py --workload simclr --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/simclr256.json 2> $O/simclr256.err; echo "rc=$?" >> $O/simclr256.err
timeout 400 python bench.py --workload clip16 --batch 1024 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/clip16_1024.json 2> $O/clip16_1024.err; echo "rc=$?" >> $O/clip16_1024.err
timeout 400 python bench.py --workload clip16 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/clip16_256.json 2> $O/clip16_256.err; echo "rc=$?" >> $O/clip16_256.err
cp gpurun_out/parity_* $O/ 2>/dev/null
tail -n 8 $O/all.log; for f in simclr512 simclr256 clip16_1024 clip16_256; do echo $f; head -c 300 $O/$f.json; echo; tail -n 2 $O/$f.err; done
