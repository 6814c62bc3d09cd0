#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run3
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
timeout 900 python -m pytest tests/test_moco_gpu.py -q --tb=short > $O/moco.log 2>&1; echo "rc=$?" >> $O/moco.log
timeout 600 python -m pytest tests/test_dp_gpu.py -q --tb=short -k "rccl or moco" > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
for arm in rccl gloo plain; do
  for rep in 1 2; do
    case $arm in
      rccl) PASSL_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$rep tests/dp_worker.py moco 2>&1 | grep DP-OK | sed "s/^/$arm $rep /" >> $O/arms.log;;
      gloo) PASSL_DP_FORCE=1 PASSL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep tests/dp_worker.py moco 2>&1 | grep DP-OK | sed "s/^/$arm $rep /" >> $O/arms.log;;
      plain) python tests/dp_worker.py moco 2>&1 | grep DP-OK | sed "s/^/$arm $rep /" >> $O/arms.log;;
    esac
  done
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
cp gpurun_out/parity_moco* $O/ 2>/dev/null
for f in ops moco dp; do tail -n 4 $O/$f.log; done; cat $O/arms.log; head -c 400 $O/bench.json
