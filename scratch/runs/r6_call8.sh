#!/bin/bash
# round 6, call 8: is it hardware-queue sharing?  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c8; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
EXTRA="" run plain A=1
EXTRA="--dp-force" run dp A=1
EXTRA="--dp-force" run dp_q8 GPU_MAX_HW_QUEUES=8
EXTRA="--dp-force" run dp_q16 GPU_MAX_HW_QUEUES=16
EXTRA="" run plain_q8 GPU_MAX_HW_QUEUES=8
EXTRA="" run plain_q2 GPU_MAX_HW_QUEUES=2
EXTRA="--dp-force" run dp_q8_inproc PASSL_HW_QUEUES=8
EXTRA="--dp-force --fresh-batches 3" run dp_q8_fresh GPU_MAX_HW_QUEUES=8
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','dp','dp_q8','dp_q16','plain_q8','plain_q2','dp_q8_inproc','dp_q8_fresh','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c8/%s.json'%f) if l.startswith('{')][-1])
        print('%-24s %9.1f img/s %7.3f ms  host %6.2f ms  fresh %s' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step'], z.get('value_fresh_inputs')))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c8/%s.err'%f).read()[-300:])
PY
