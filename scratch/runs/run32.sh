#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(d['metric'][:50], d['value'], d['ms_per_step'], 'alloc', c.get('hbm_allocated_gb'), 'resv', c.get('hbm_reserved_gb'), 'retries', c.get('allocator_retries'))
"; }
timeout 900 python -m pytest tests/test_moco_gpu.py tests/test_clas_gpu.py tests/test_simclr_gpu.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | show
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 6 2>/dev/null | show
timeout 600 python bench.py --workload simclr --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 6 2>/dev/null | show
