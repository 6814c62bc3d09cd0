#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call11; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name', e); print(open('$O/$name.err').read()[-800:])
PY
}
run base A=1
run bk32_4 PASSL_OPTIONS=igemm_ring_bk=32
run bk32_3 PASSL_OPTIONS=igemm_ring_bk=32,igemm_ring_stages32=3
run bm256 PASSL_OPTIONS=igemm_ring_bm=256
run bnu8 PASSL_OPTIONS=bn_stream_unroll=8
run bnu2 PASSL_OPTIONS=bn_stream_unroll=2
run sidebatch8 PASSL_SIDE_BATCH=8
run sidebatch2 PASSL_SIDE_BATCH=2
run base2 A=1
