#!/bin/bash
# round 6, call 11: WHEN the communicator is created
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c11; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp_eager_init A=1
EXTRA="--dp-force" run dp_lazy_at_paramsync PASSL_DIST_LAZY=1
EXTRA="--dp-force" run dp_lazy_at_step1_gather PASSL_DIST_LAZY=1 PASSL_DP_DIAG=noparamsync
EXTRA="--dp-force" run dp_lazy_at_step1_bucket PASSL_DIST_LAZY=1 PASSL_DP_DIAG=noparamsync,nogather
EXTRA="--dp-force" run dp_eager_noparamsync PASSL_DP_DIAG=noparamsync
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','dp_eager_init','dp_lazy_at_paramsync','dp_lazy_at_step1_gather','dp_lazy_at_step1_bucket','dp_eager_noparamsync','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c11/%s.json'%f) if l.startswith('{')][-1])
        d=z.get('dist') or {}
        print('%-26s %9.1f img/s %7.3f ms  host %6.2f ms  reserved %s GB' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'], z['config']['hbm_reserved_gb']))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c11/%s.err'%f).read()[-300:])
PY
