#!/bin/bash
# round-6 evidence run: everything that goes under profiles/r06_* (summaries are made on the box; the rocpd databases stay
# there).  PMC passes and the serial trace use --eager so that the number of steps in the trace is exactly warm-up + steps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_evidence
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
if [ "$1" != "nosuite" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
  python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
fi
{
  echo "== tools/kbench check"; timeout 120 tools/kbench check | tail -3
  echo "== tools/kbench check igemm_persist=1 igemm_persist_grid=8 (opt-in persistent form)"; timeout 120 tools/kbench check igemm_persist=1 igemm_persist_grid=8 | tail -2
  echo "== tools/kbench check conv3x3_wave_rows=8"; timeout 120 tools/kbench check conv3x3_wave_rows=8 | tail -2
  echo "== tools/kbench wcheck"; timeout 120 tools/kbench wcheck | tail -2
  echo "== tools/kbench time (forward + statistics)"; timeout 60 tools/kbench time
  echo "== tools/kbench ab conv3x3_wave=0,1"; timeout 60 tools/kbench ab conv3x3_wave=0,1 | head -4
  echo "== tools/kbench vtime"; timeout 60 tools/kbench vtime
} > $O/kbench.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --eager --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --eager --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r06_pmc_traffic.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o m -- $B --eager --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $T/pmc_mfma_summary.py $(db /tmp/p_mfma) 3 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --no-cpu-baseline --no-kernel-timing --eager --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 4 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 12 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 4 (12 steps in the trace: 3 eager, 1 recording, 8 replayed from the native step plan)" > $O/kernel_stats.txt 2>&1
PASSL_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof0 -o s -- $B --eager --steps 8 --warmup 2 > $O/prof0.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof0) 10 "PASSL_OVERLAP=0 (no side stream: every kernel's duration is its own) rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --eager --steps 8 --warmup 2" > $O/kernel_stats_serial.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$CSV" ]; then
  timeout 120 python $T/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
  timeout 120 python $T/trace_chain.py $CSV 4 > $O/trace_chain.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --fresh-batches 3 > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-force > $O/bench_moco_dp_forced.json 2> $O/bench_moco_dp_forced.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > $O/bench_moco_plain_after_dp.json 2> $O/bench_moco_plain_after_dp.err
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype fp32 > $O/bench_moco_fp32.json 2> $O/bench_moco_fp32.err
timeout 300 python bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --no-kernel-timing > $O/bench_moco_eager.json 2> $O/bench_moco_eager.err
for w in simclr mae clip16 linprobe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
done
timeout 600 python bench.py --workload simclr --batch 512 --no-cpu-baseline --steps 10 --warmup 6 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 600 python bench.py --workload clip16 --batch 1024 --no-cpu-baseline --steps 6 --warmup 5 >> $O/bench_workloads.jsonl 2>> $O/bench_workloads.err
timeout 300 python scratch/bench_convs.py > $O/conv_layers.txt 2>&1
tail -4 $O/tests_gpu.log; tail -2 $O/smoke.log; head -c 700 $O/bench_moco.json; echo; head -14 $O/kernel_stats.txt; cut -c1-160 $O/bench_workloads.jsonl
