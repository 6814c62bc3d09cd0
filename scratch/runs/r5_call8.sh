#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call8; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name', e); print(open('$O/$name.err').read()[-800:])
PY
}
run base A=1
run stem_side PASSL_STEM_WGRAD_MAIN=0
run p8_nk4 PASSL_OPTIONS=igemm_8p_min_nk=4
run p8_nk4_m2 PASSL_OPTIONS=igemm_8p_min_nk=4,igemm_8p=2
run ring_nk4 PASSL_OPTIONS=igemm_ring_min_nk=4
run lean0 PASSL_IGEMM_LEAN=0
run base2 A=1
