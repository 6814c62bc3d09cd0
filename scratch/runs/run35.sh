#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in moco clip mae clip16; do python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 6 2>/dev/null | cut -c1-150; done
