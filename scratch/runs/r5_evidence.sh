#!/bin/bash
# round-5 evidence run: everything that goes under profiles/r05_* (summaries are made on the box; the rocpd databases
# stay there — gpurun_out/ is capped at 64 MiB).  PMC passes and the serial trace use --eager so that the number of
# steps in the trace is exactly warm-up + steps; the overlapped traces are the default product path (native step plan:
# 3 eager steps + 1 recording step inside --warmup 4, then replays).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_evidence
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
if [ "$1" != "nosuite" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
  python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
fi
{
  echo "== tools/kbench check"; timeout 60 tools/kbench check | tail -2
  echo "== tools/kbench wcheck"; timeout 120 tools/kbench wcheck | tail -2
  echo "== tools/kbench wtime (weight gradients: default selection, then wgrad_halo=0)"; timeout 60 tools/kbench wtime; timeout 60 tools/kbench wtime wgrad_halo=0 rows=6
  echo "== tools/kbench fincheck"; timeout 120 tools/kbench fincheck
  echo "== tools/kbench fintime"; timeout 90 tools/kbench fintime
  echo "== tools/kbench finstress"; timeout 200 tools/kbench finstress iters=45
  echo "== tools/kbench time (forward + statistics)"; timeout 60 tools/kbench time
} > $O/kbench.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --eager --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --eager --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r05_pmc_traffic.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o m -- $B --eager --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $T/pmc_mfma_summary.py $(db /tmp/p_mfma) 3 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --no-cpu-baseline --no-kernel-timing --eager --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 4 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 12 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 4 (12 steps in the trace: 3 eager, 1 recording, 8 replayed from the native step plan)" > $O/kernel_stats.txt 2>&1
PASSL_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof0 -o s -- $B --eager --steps 8 --warmup 2 > $O/prof0.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof0) 10 "PASSL_OVERLAP=0 (no side stream: every kernel's duration is its own) rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --eager --steps 8 --warmup 2" > $O/kernel_stats_serial.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
python $T/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
python $T/trace_chain.py $CSV 4 > $O/trace_chain.txt 2>&1
for w in mae clip16; do
  PASSL_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o s -- $B --eager --workload $w --steps 4 --warmup 2 > $O/prof_$w.log 2>&1
  python $T/rocpd_summary.py $(db /tmp/p_$w) 6 "PASSL_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --eager --steps 4 --warmup 2 (6 steps in the trace)" > $O/kernel_stats_${w}_serial.txt 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
timeout 300 python bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --no-kernel-timing > $O/bench_moco_eager.json 2> $O/bench_moco_eager.err
for w in simclr mae clip clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
tail -4 $O/tests_gpu.log; tail -2 $O/smoke.log; head -c 900 $O/bench_moco.json; echo; head -14 $O/kernel_stats.txt; cat $O/trace_timeline.txt | head -12; cut -c1-200 $O/bench_workloads.jsonl
