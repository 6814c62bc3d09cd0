#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_moco_gpu.py -q -x 2>&1 | tail -3
for f in 0 1 0 1; do echo "own stream $f"; PASSL_FORK_OWN_STREAM=$f python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | cut -c1-150; done
