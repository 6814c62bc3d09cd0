#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3j
export HSA_ENABLE_IPC_MODE_LEGACY=0
PASSL_DIST_BACKEND=gloo PASSL_DEVICE_INDEX=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 tests/dp_worker.py simsiam > gpurun_out/r3j/dp_simsiam.log 2>&1
echo "exit $?" >> gpurun_out/r3j/dp_simsiam.log
grep -v "^\[Gloo\]\|^W0\|^$\|amdgpu" gpurun_out/r3j/dp_simsiam.log | tail -25
