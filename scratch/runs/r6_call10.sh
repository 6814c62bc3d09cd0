#!/bin/bash
# round 6, call 10: streams created before the communicator (default now) vs after
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c10; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp A=1
EXTRA="--dp-force" run dp_nopre PASSL_PRECREATE_STREAMS=0
EXTRA="--dp-force" run dp_lazy PASSL_DIST_LAZY=1
EXTRA="--dp-force" run dp_lazy_nopre PASSL_DIST_LAZY=1 PASSL_PRECREATE_STREAMS=0
EXTRA="" run plain_nopre PASSL_PRECREATE_STREAMS=0
EXTRA="--dp-force" run dp_dry PASSL_DP_DRYRUN=1
EXTRA="--dp-force" run dp_bf16wire PASSL_DP_WIRE=bf16
EXTRA="--dp-force" run dp2 A=1
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','dp','dp_nopre','dp_lazy','dp_lazy_nopre','plain_nopre','dp_dry','dp_bf16wire','dp2','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c10/%s.json'%f) if l.startswith('{')][-1])
        d=z.get('dist') or {}
        print('%-18s %9.1f img/s %7.3f ms  host %6.2f ms  exposed %s' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'], d.get('allreduce_exposed_ms')))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c10/%s.err'%f).read()[-300:])
PY
