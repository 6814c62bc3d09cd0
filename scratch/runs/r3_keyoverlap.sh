#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_moco_gpu.py -m gpu -q -x > gpurun_out/r3k/tests_moco.log 2>&1; echo "exit $?" >> gpurun_out/r3k/tests_moco.log
for v in 0 1 0 1; do
  echo "PASSL_KEY_OVERLAP=$v $(PASSL_KEY_OVERLAP=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | cut -c1-160)" >> gpurun_out/r3k/ab.txt
done
tail -4 gpurun_out/r3k/tests_moco.log; cat gpurun_out/r3k/ab.txt
