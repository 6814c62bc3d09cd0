#!/bin/bash
# round-3 evidence run: everything that goes under profiles/r03_* (summaries are made on the box; the
# rocpd databases stay there — gpurun_out/ is capped at 64 MiB)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_evidence
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
# 1. PMC passes first: bench.py's roofline.traffic reads the committed summary of THIS command
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r03_pmc_traffic.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o m -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $T/pmc_mfma_summary.py $(db /tmp/p_mfma) 3 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
# 2. kernel trace (default = with the side stream; serial = every kernel's duration is its own)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 2 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 10 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 (10 steps in the trace)" > $O/kernel_stats.txt 2>&1
PASSL_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof0 -o s -- $B --steps 8 --warmup 2 > $O/prof0.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof0) 10 "PASSL_OVERLAP=0 (no side stream: every kernel's duration is its own) rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2" > $O/kernel_stats_serial.txt 2>&1
for w in mae clip16; do
  PASSL_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o s -- $B --workload $w --steps 4 --warmup 2 > $O/prof_$w.log 2>&1
  python $T/rocpd_summary.py $(db /tmp/p_$w) 6 "PASSL_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 (6 steps in the trace)" > $O/kernel_stats_${w}_serial.txt 2>&1
done
cd $GRAFT_REPO_ROOT
# 3. the bench line itself (default flags)
timeout 600 python bench.py > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
# 4. the other workloads
for w in simclr mae clip clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
timeout 100 python scratch/bench_attn.py > $O/attn.txt 2>&1
head -c 700 $O/bench_moco.json; echo; head -14 $O/kernel_stats.txt; cut -c1-200 $O/bench_workloads.jsonl
