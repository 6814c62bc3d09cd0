#!/bin/bash
# round 6, call 5: what the forced data-parallel path costs, by part (no profiler: it makes the host the bottleneck)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp A=1
EXTRA="--dp-force" run dp_dry PASSL_DP_DRYRUN=1
EXTRA="--dp-force" run dp_thread PASSL_DP_THREAD=1
EXTRA="--dp-force --dp-buckets 1" run dp_1bucket A=1
EXTRA="--dp-force --dp-buckets 8" run dp_8buckets A=1
EXTRA="--dp-force --eager" run dp_eager A=1
EXTRA="--eager" run plain_eager A=1
EXTRA="" run plain2 A=1
EXTRA="--dp-force" run dp2 A=1
python - <<'PY'
import json, glob, os
for f in ['plain','dp','dp_dry','dp_thread','dp_1bucket','dp_8buckets','dp_eager','plain_eager','plain2','dp2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c5/%s.json'%f) if l.startswith('{')][-1])
        d=z.get('dist') or {}
        print('%-12s %9.1f img/s %7.3f ms  host %6.2f ms  segs %s  coll_host %s ms / %s calls  exposed %s' % (
            f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'],
            z['config']['step_launch'].split('stream waits on')[-1][:24].strip() if 'segment' in z['config']['step_launch'] else 'eager',
            d.get('collective_host_ms_per_step'), d.get('collective_host_calls_per_step'), d.get('allreduce_exposed_ms')))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c5/%s.err'%f).read()[-400:])
PY
