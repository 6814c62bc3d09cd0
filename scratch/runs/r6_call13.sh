#!/bin/bash
# round 6, call 13: two-bucket default of the data-parallel path (+ issuer thread), DP test suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c13; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp A=1
EXTRA="--dp-force" run dp_thread PASSL_DP_THREAD=1
EXTRA="--dp-force --dp-buckets 4" run dp_4buckets A=1
EXTRA="--dp-force --dp-buckets 4" run dp_4buckets_thread PASSL_DP_THREAD=1
EXTRA="--dp-force" run dp_tail256k PASSL_DP_TAIL_ELEMS=262144
EXTRA="" run plain2 A=1
EXTRA="--dp-force" run dp2 A=1
python - <<'PY'
import json
for f in ['plain','dp','dp_thread','dp_4buckets','dp_4buckets_thread','dp_tail256k','plain2','dp2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c13/%s.json'%f) if l.startswith('{')][-1])
        d=z.get('dist') or {}
        print('%-22s %9.1f img/s %7.3f ms  host %6.2f ms  buckets %s exposed %s' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'], d.get('grad_buckets'), d.get('allreduce_exposed_ms')))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c13/%s.err'%f).read()[-300:])
PY
( timeout 900 python -m pytest tests/test_dp_gpu.py -q -x 2>&1 | tail -5 ) > $O/tests_dp.log 2>&1
( PASSL_DP_THREAD=1 timeout 600 python -m pytest tests/test_dp_gpu.py -q -x -k "rccl or two_ranks_one_gpu" 2>&1 | tail -5 ) > $O/tests_dp_thread.log 2>&1
tail -3 $O/tests_dp.log $O/tests_dp_thread.log
