#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(d['metric'][:50], d['value'], d['ms_per_step'], 'alloc', c.get('hbm_allocated_gb'), 'resv', c.get('hbm_reserved_gb'), 'retries', c.get('allocator_retries'))
"; }
echo "== default allocator"
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 6 2>/dev/null | show
echo "== expandable_segments"
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 6 2>&1 | grep -v "^\[" | show
echo "== overlap off"
PASSL_OVERLAP=0 timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 6 2>/dev/null | show
echo "== bs384"
timeout 600 python bench.py --workload simclr --batch 384 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 6 2>/dev/null | show
