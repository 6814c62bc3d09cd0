#!/bin/bash
# in-step A/B of kernel options that were chosen on stand-alone timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/c29; rm -rf $O; mkdir -p $O
run() { # label opts rep
  PASSL_OPTIONS=$2 timeout 300 python bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 (PASSL_OPTIONS=$2) rep $3: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2 3; do
  run default "" $rep
  run wgrad_pipe_32x4 wgrad_pipe=1 $rep
  run wave_rows8 conv3x3_wave_rows=8 $rep
  run halo_stages3 wgrad_halo_stages=3 $rep
  run both wgrad_pipe=1,conv3x3_wave_rows=8 $rep
done | tee $O/ab.txt
