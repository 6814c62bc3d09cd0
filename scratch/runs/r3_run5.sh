#!/bin/bash
# round 3, run 5: atomic-free reductions (ViT / CLIP / linear probe), MoCo-v3 parity, 2-rank MoCo-v3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_mae_gpu.py tests/test_clip_gpu.py tests/test_clas_gpu.py tests/test_mocov3_gpu.py tests/test_ops_gpu.py -m gpu -q > gpurun_out/r3e/tests.log 2>&1
echo "exit $?" >> gpurun_out/r3e/tests.log
PASSL_DIST_BACKEND=gloo PASSL_DEVICE_INDEX=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tests/dp_worker.py mocov3 > gpurun_out/r3e/dp_mocov3.log 2>&1
echo "exit $?" >> gpurun_out/r3e/dp_mocov3.log
tail -40 gpurun_out/r3e/tests.log; grep -v "^\[Gloo\]\|^W0\|^$" gpurun_out/r3e/dp_mocov3.log | tail -30
