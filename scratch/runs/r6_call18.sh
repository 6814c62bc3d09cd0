#!/bin/bash
# latest wave-kernel edit (weights staged under the first halo request): check + time; DP tests after the bench barrier change;
# the SimSiam small bf16 golden with the wave kernel on / off (is the step-0 predictor.3.bias deviation the kernel's statistics rounding?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c18; rm -rf $O; mkdir -p $O
K=tools/kbench
( echo "== check"; timeout 200 $K check | tail -5
  echo "== wave vs ring"; timeout 60 $K ab conv3x3_wave=0,1 | sed -n 1,6p
  echo "== rows 8 vs 4"; timeout 60 $K ab conv3x3_wave_rows=8,4 | sed -n 1,6p
) > $O/kbench.txt 2>&1
for w in 1 0; do
  PASSL_CONV3X3_WAVE=$w timeout 600 python -m pytest tests/test_simsiam_gpu.py -q -m gpu -k "golden_small" > $O/simsiam_wave$w.log 2>&1
  cp gpurun_out/parity_simsiam_r50_small_bfloat16.txt $O/parity_simsiam_small_bf16_wave$w.txt
done
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_kbench_gpu.py "tests/test_ops_gpu.py" -q -m gpu -k "dp or kbench or wave" > $O/tests_misc.log 2>&1
tail -3 $O/*.log; cat $O/kbench.txt; grep -h "predictor.3.bias" $O/parity*.txt

# what is live between two steps (SimCLR bs 512 ran out of memory while the plan recorded)
timeout 300 python scratch/probe/mem_probe.py 64 > $O/mem_probe_bs64.txt 2>&1
tail -40 $O/mem_probe_bs64.txt
timeout 900 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_simclr_bs512.json 2> $O/bench_simclr_bs512.err
tail -3 $O/bench_simclr_bs512.err | cut -c1-600; grep '^{' $O/bench_simclr_bs512.json | cut -c1-300
