#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call14; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_kbench_gpu.py tests/test_dp_gpu.py -m gpu -x -q -k "kbench or c_abi or finalize or bf16_gradient_wire or two_ranks_one_gpu" > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -6 $O/tests.log
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
for i in 1 2; do timeout 300 $B > $O/bench$i.json 2> $O/bench$i.err; python - <<PY
import json
d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench$i', d['value'], d['ms_per_step'])
PY
done
