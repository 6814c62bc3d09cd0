#!/bin/bash
# First GPU call of the next round (~20 s of GPU time, no Python): the opt-in kernels that round 4 left checked by
# the host emulators only, bit for bit, and their timings against the product kernels.
#   /usr/local/graft/bin/gpurun --timeout 120 -- 'bash scratch/runs/r5_first_call.sh'
O=gpurun_out/r5_first; mkdir -p $O
K=tools/kbench
[ -x $K ] || bash tools/build_kbench.sh
{
  echo "== persistent weights-resident 3x3 kernel (igemm_halo=2): check"; timeout 40 $K check igemm_halo=2 igemm_halo_max_c=512
  echo "== forward 3x3: product | halo 2-D | persistent"
  KBENCH_STAMPS=1 timeout 40 $K sweep igemm_halo=0 igemm_halo=1,igemm_halo_max_c=512 igemm_halo=2,igemm_halo_max_c=512
  echo "== BatchNorm: finalize inside the streaming kernels against the separate launches"; timeout 60 $K bncheck; timeout 30 $K bntime
  echo "== weight gradient: overhanging patches (wgrad_halo=2): check"; timeout 60 $K wcheck wgrad_halo=2
  echo "== weight gradient timings: product, then wgrad_halo=2 with its own slice counts (512 / blocks)"
  timeout 20 $K wtime rows=6
  for row in "1 512" "2 128" "3 32" "4 8"; do set -- $row; timeout 10 $K wtime rows=$1 wgrad_halo=2 splits=$2 | tail -2 | head -1; done
} > $O/kbench.txt 2>&1
tail -60 $O/kbench.txt
# then, with the suite:  PASSL_WGRAD_HALO=2 PASSL_BN_FUSED_FINALIZE=1 python -m pytest tests -m gpu -x -q   (the switches
# for the defaults), and python bench.py --steps 20 --warmup 5 with and without them
