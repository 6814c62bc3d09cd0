#!/bin/bash
# SimCLR at the BASELINE per-GPU batch (512) with the bench fix; SimSiam golden with the new floor; the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_simclr_bs512.json 2> $O/bench_simclr_bs512.err
tail -3 $O/bench_simclr_bs512.err | cut -c1-400; grep '^{' $O/bench_simclr_bs512.json | cut -c1-400
timeout 600 python -m pytest tests/test_simsiam_gpu.py -q -m gpu > $O/simsiam.log 2>&1; tail -3 $O/simsiam.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
grep '^{' $O/bench_driver_cmd.json | cut -c1-300; tail -4 $O/bench_driver_cmd.err
