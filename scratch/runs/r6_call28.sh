#!/bin/bash
# smaller weight-gradient tiles (166 instead of 236 VGPRs: two main-chain waves fit beside one) at the same workgroup count
cd $GRAFT_REPO_ROOT
O=gpurun_out/c28; rm -rf $O; mkdir -p $O
run() { # label opts pipe halo rep
  PASSL_OPTIONS=$2 PASSL_WGRAD_TARGET_BLOCKS=$3 PASSL_WGRAD_HALO_TARGET_BLOCKS=$4 timeout 300 python bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moco $1 (opts=$2 pipe $3 halo $4) rep $5: %.3f ms' % d['ms_per_step'])"
}
for rep in 1 2 3; do
  run default "" 0 0 $rep
  run tile128x64_t128 wgrad_tile=3 128 0 $rep
  run tile64x128_t128 wgrad_tile=2 128 0 $rep
  run tile128x64_t192 wgrad_tile=3 192 0 $rep
  run tile64x64_t64 wgrad_tile=1 64 0 $rep
done | tee $O/ab.txt
