#!/bin/bash
# round-2 evidence run: everything that goes under profiles/r02_*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run13
mkdir -p $O
timeout 600 python bench.py > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
for w in simclr mae clip clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 400 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
timeout 300 python scratch/count_torch_ops.py > $O/aten_ops.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o s -- $B --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o f -- $B --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_write -o w -- $B --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o m -- $B --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
head -c 1500 $O/bench_moco.json; echo; cut -c1-200 $O/bench_workloads.jsonl
