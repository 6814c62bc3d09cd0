#!/bin/bash
# round 6, call 7: merely creating the world-1 RCCL communicator slows the step by ~8 % — what in it?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c7; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B ${EXTRA} > $O/$name.json 2> $O/$name.err; }
D="PASSL_DP_DRYRUN=1 PASSL_DP_DIAG=noreducer,nogather"
EXTRA="" run plain A=1
EXTRA="--dp-force" run init_nccl $D
EXTRA="--dp-force" run init_nccl_lazy $D PASSL_DIST_LAZY=1
EXTRA="--dp-force" run init_gloo $D PASSL_DIST_BACKEND=gloo
EXTRA="--dp-force" run init_nccl_nomonitor $D TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
EXTRA="--dp-force" run init_nccl_1chan $D NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
EXTRA="--dp-force" run init_nccl_sdma $D HSA_ENABLE_SDMA=0
EXTRA="--dp-force" run init_nccl_legacy $D HSA_ENABLE_IPC_MODE_LEGACY=1
EXTRA="" run plain_rcclpreload LD_PRELOAD=/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so
EXTRA="" run plain2 A=1
python - <<'PY'
import json
for f in ['plain','init_nccl','init_nccl_lazy','init_gloo','init_nccl_nomonitor','init_nccl_1chan','init_nccl_sdma','init_nccl_legacy','plain_rcclpreload','plain2']:
    try:
        z=json.loads([l for l in open('gpurun_out/c7/%s.json'%f) if l.startswith('{')][-1])
        print('%-24s %9.1f img/s %7.3f ms  host %6.2f ms' % (f, z['value'], z['ms_per_step'], z['config']['host_enqueue_ms_per_step']))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c7/%s.err'%f).read()[-300:])
PY
