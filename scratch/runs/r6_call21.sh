#!/bin/bash
# final-tree bench lines: default command with moving inputs, forced data-parallel and plain on the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/c21; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --fresh-batches 3 > $O/bench_moco.json 2> $O/bench_moco.err
timeout 300 python bench.py --steps 20 --warmup 5 --dp-force --no-cpu-baseline > $O/bench_moco_dp_forced.json 2> $O/bench_moco_dp_forced.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_moco_plain.json 2> $O/bench_moco_plain.err
for f in bench_moco bench_moco_dp_forced bench_moco_plain; do grep '^{' $O/$f.json | cut -c1-260; done
