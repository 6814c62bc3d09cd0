#!/bin/bash
# does the forced data-parallel step lose concurrency to hardware-queue sharing?  (4 product streams + RCCL's own against GPU_MAX_HW_QUEUES)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c31; rm -rf $O; mkdir -p $O
run() { # label dpflag queues rep
  if [ "$3" = "default" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$3; fi
  timeout 300 python bench.py $2 --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 GPU_MAX_HW_QUEUES=$3 rep $4: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2; do
  for q in default 2 3 5 6 8; do run plain "" $q $rep; run dp --dp-force $q $rep; done
done | tee $O/ab.txt
