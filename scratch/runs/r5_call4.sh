#!/bin/bash
# step time with the round-5 defaults (wgrad_halo=2, one-launch finalize), A/B against wgrad_halo=0, key GPU tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call4; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
timeout 300 $B > $O/bench_new.json 2> $O/bench_new.err
PASSL_WGRAD_HALO=0 timeout 300 $B > $O/bench_nohalo.json 2> $O/bench_nohalo.err
timeout 300 $B > $O/bench_new2.json 2>> $O/bench_new.err
timeout 900 python -m pytest tests/test_kbench_gpu.py tests/test_ops_gpu.py tests/test_layers_gpu.py tests/test_moco_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
for f in bench_new bench_nohalo bench_new2; do python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'])
PY
done
tail -5 $O/tests.log
