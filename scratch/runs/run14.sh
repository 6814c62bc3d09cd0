#!/bin/bash
# round-2 evidence run: everything that goes under profiles/r02_* (summaries are made on the box; the
# rocpd databases stay there — gpurun_out/ is capped at 64 MiB)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_run14
mkdir -p $O
timeout 600 python bench.py > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
timeout 300 python scratch/count_torch_ops.py > $O/aten_ops.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 2 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 10 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace)" > $O/kernel_stats.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o m -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $T/pmc_mfma_summary.py $(db /tmp/p_mfma) 3 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
cd $GRAFT_REPO_ROOT
head -c 3000 $O/bench_moco.json; echo; head -40 $O/kernel_stats.txt; head -30 $O/pmc_mfma.txt; head -30 $O/pmc_traffic.txt
