#!/bin/bash
# round 6, call 12: the remaining 0.6 ms of the (lazy-communicator) data-parallel path: machinery or communicator?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c12; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp A=1
EXTRA="--dp-force" run machinery_only PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noparamsync,nogather
EXTRA="--dp-force" run machinery_noedges PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noparamsync,nogather,noedges
EXTRA="--dp-force" run dp_1bucket A=1
EXTRA="--dp-force --dp-buckets 1" run dp_1bucketb A=1
EXTRA="--dp-force" run dp_nogather PASSL_DP_DIAG=nogather
EXTRA="--dp-force" run dp_thread PASSL_DP_THREAD=1
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','dp','machinery_only','machinery_noedges','dp_1bucketb','dp_nogather','dp_thread','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c12/%s.json'%f) if l.startswith('{')][-1])
        d=z.get('dist') or {}
        print('%-22s %9.1f img/s %7.3f ms  host %6.2f ms  %s' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'], z['config']['step_launch'][60:200]))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c12/%s.err'%f).read()[-300:])
PY
