#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/graph_probe.log; : > $O
for env in "PASSL_OVERLAP=1"; do
  for c in step; do
    echo "=== $env $c" >> $O
    env $env timeout 120 python scratch/graph_probe.py $c 2>&1 | grep -v amdgpu.ids | tail -4 >> $O
  done
done
cat $O
timeout 600 python -m pytest tests/test_moco_gpu.py -x -q -m gpu -k "step_graph" 2>&1 | tail -5
for g in 1 0; do PASSL_GRAPH=$g timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 6 --no-kernel-timing 2>/dev/null | cut -c1-230; done
for w in clip mae clip16; do for g in 1 0; do PASSL_GRAPH=$g timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 --no-kernel-timing 2>&1 | tail -1 | cut -c1-200; done; done
