#!/bin/bash
# round 6, call 9: the RCCL communicator's existence costs ~8 % of the step — knobs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c9; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
D="PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noreducer,nogather"
EXTRA="" run plain A=1
EXTRA="--dp-force" run init $D
EXTRA="--dp-force" run init_q1 $D GPU_MAX_HW_QUEUES=1
EXTRA="--dp-force" run init_q2 $D GPU_MAX_HW_QUEUES=2
EXTRA="--dp-force" run dp_lazy_real PASSL_DIST_LAZY=1
EXTRA="--dp-force" run init_noaffinity $D NCCL_IGNORE_CPU_AFFINITY=1
EXTRA="--dp-force" run init_nointerrupt $D HSA_ENABLE_INTERRUPT=0
EXTRA="--dp-force" run init_nomscc $D RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
EXTRA="--dp-force" run init_smallbuf $D NCCL_BUFFSIZE=65536 NCCL_MAX_NCHANNELS=2
EXTRA="--dp-force" run init_devkernarg $D HIP_FORCE_DEV_KERNARG=1
EXTRA="--dp-force" run init_nodirect $D AMD_DIRECT_DISPATCH=0
EXTRA="--dp-force" run init_debugver $D NCCL_DEBUG=INFO
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','init','init_q1','init_q2','dp_lazy_real','init_noaffinity','init_nointerrupt','init_nomscc','init_smallbuf','init_devkernarg','init_nodirect','init_debugver','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c9/%s.json'%f) if l.startswith('{')][-1])
        print('%-24s %9.1f img/s %7.3f ms  host %6.2f ms' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step']))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c9/%s.err'%f).read()[-300:])
PY
grep -i "stack\|scratch\|limit\|affinity\|stream" $O/init_debugver.err | head -30
