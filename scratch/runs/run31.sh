#!/bin/bash
# round-2 evidence run: everything that goes under profiles/r02_* (summaries are made on the box; the
# rocpd databases stay there — gpurun_out/ is capped at 64 MiB)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_run31
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
# 1. PMC passes first: bench.py's roofline.traffic reads the committed summary of THIS command
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
mkdir -p $GRAFT_REPO_ROOT/profiles; cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r02_pmc_traffic.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o m -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $T/pmc_mfma_summary.py $(db /tmp/p_mfma) 3 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d /tmp/p_stall -o st -- $B --steps 1 --warmup 1 > $O/pmc_stall.log 2>&1
python $T/pmc_stall_summary.py $(db /tmp/p_stall) > $O/pmc_stall.txt 2>&1
# 2. kernel trace
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 2 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 10 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace)" > $O/kernel_stats.txt 2>&1
PASSL_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof0 -o s -- $B --steps 8 --warmup 2 > $O/prof0.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof0) 10 "PASSL_OVERLAP=0 (no side stream: every kernel's duration is its own) rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2" > $O/kernel_stats_serial.txt 2>&1
cd $GRAFT_REPO_ROOT
# 3. the bench line itself (default flags) + A/B of the stream options on this box
timeout 600 python bench.py > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
for v in "PASSL_OVERLAP=0" "PASSL_FORK_DOWNSAMPLE=0" "PASSL_BN_STREAM_UNROLL=0" "PASSL_STEM_KERNEL=0" "X=1"; do
  echo "$v: $(env $v python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 6 2>/dev/null | cut -c1-140)" >> $O/ab.txt
done
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
timeout 300 python scratch/count_torch_ops.py > $O/aten_ops.txt 2>&1
timeout 100 python scratch/bench_bn.py > $O/bn_stream.txt 2>&1
timeout 100 python scratch/hbm_ceiling.py > $O/hbm_ceiling.txt 2>&1
timeout 100 python scratch/bench_stem.py > $O/stem.txt 2>&1
timeout 100 python scratch/bench_pool.py > $O/pool.txt 2>&1
# 4. the other workloads
for w in simclr mae clip clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 600 python bench.py --workload clip16 --batch 1024 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
head -c 600 $O/bench_moco.json; echo; cat $O/ab.txt; head -12 $O/kernel_stats.txt
