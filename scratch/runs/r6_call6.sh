#!/bin/bash
# round 6, call 6: bisect the forced data-parallel overhead (dry run = no collective library calls)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c6; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp_dry PASSL_DP_DRYRUN=1
EXTRA="--dp-force" run dry_noedges PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noedges
EXTRA="--dp-force" run dry_nogather PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=nogather
EXTRA="--dp-force" run dry_noreducer PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noreducer
EXTRA="--dp-force" run dry_noreducer_nogather PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noreducer,nogather
EXTRA="--dp-force" run dry_noedges_nogather PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noedges,nogather
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','dp_dry','dry_noedges','dry_nogather','dry_noreducer','dry_noreducer_nogather','dry_noedges_nogather','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c6/%s.json'%f) if l.startswith('{')][-1])
        print('%-24s %9.1f img/s %7.3f ms  host %6.2f ms' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step']))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c6/%s.err'%f).read()[-300:])
PY
