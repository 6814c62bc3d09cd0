#!/bin/bash
# round 3: throughput of the v2 linear-probe recipes (and the pre-training recipes after the schedule fix)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3_lp_bench
rm -rf $O; mkdir -p $O
B="timeout 120 python scratch/bench_v2.py"
{
$B configs/v2/simsiam_resnet50_lp_synthetic.yaml 512 fp32 20
$B configs/v2/simsiam_resnet50_lp_synthetic.yaml 512 bf16 20
$B configs/v2/mocov3_vit_base_lp_synthetic.yaml 128 bf16 20
$B configs/v2/mocov3_vit_base_lp_synthetic.yaml 512 bf16 20
$B configs/v2/mocov3_vit_base_pt_synthetic.yaml 128 bf16 20
} 2>&1 | grep -v amdgpu.ids | grep "^{" > $O/bench.jsonl
cat $O/bench.jsonl
