#!/bin/bash
# round 6, call 17: wave kernel inside the product: conv / layer / MoCo parity tests, step A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c17; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_layers_gpu.py tests/test_moco_gpu.py tests/test_kbench_gpu.py tests/test_step_plan_gpu.py -q -x 2>&1 | tail -12 ) > $O/tests.log 2>&1
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; }
run wave1 A=1
run wave0 PASSL_CONV3X3_WAVE=0
run wave1_rows8 PASSL_OPTIONS=conv3x3_wave_rows=8
run wave1b A=1
run wave0b PASSL_CONV3X3_WAVE=0
tail -4 $O/tests.log
python - <<'PY'
import json
for f in ['wave1','wave0','wave1_rows8','wave1b','wave0b']:
    try:
        z=json.loads([l for l in open('gpurun_out/c17/%s.json'%f) if l.startswith('{')][-1])
        print('%-14s %9.1f img/s %7.3f ms  loss %s' % (f, z['value'], z['ms_per_step'], z['config']['final_loss']))
    except Exception as e:
        print(f, 'FAILED', e, open('gpurun_out/c17/%s.err'%f).read()[-400:])
PY
