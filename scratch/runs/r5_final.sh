#!/bin/bash
# final run of round 5 (after the bnb2 change): full GPU suite, smoke, the driver's bench command, kernel statistics
# (replayed plan + serial), timeline / chain, SimCLR line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_final
rm -rf $O; mkdir -p $O
T=$GRAFT_REPO_ROOT/tools
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "exit $?" >> $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof -o s -- $B --steps 8 --warmup 4 > $O/prof.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof) 12 "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 4 (12 steps in the trace: 3 eager, 1 recording, 8 replayed from the native step plan)" > $O/kernel_stats.txt 2>&1
PASSL_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_prof0 -o s -- $B --eager --steps 8 --warmup 2 > $O/prof0.log 2>&1
python $T/rocpd_summary.py $(db /tmp/p_prof0) 10 "PASSL_OVERLAP=0 (no side stream: every kernel's duration is its own) rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --eager --steps 8 --warmup 2" > $O/kernel_stats_serial.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_csv -o t -- $B --steps 8 --warmup 4 > $O/prof_csv.log 2>&1
CSV=$(ls /tmp/p_csv/*/*kernel_trace.csv /tmp/p_csv/*kernel_trace.csv 2>/dev/null | head -1)
python $T/trace_timeline.py $CSV 6 > $O/trace_timeline.txt 2>&1
python $T/trace_chain.py $CSV 4 > $O/trace_chain.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $B --eager --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- $B --eager --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $T/pmc_summary.py $(db /tmp/p_fetch) $(db /tmp/p_write) 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r05_pmc_traffic.json
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_moco.json 2> $O/bench_moco.err; echo "rc=$?" >> $O/bench_moco.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_moco2.json 2> $O/bench_moco2.err
timeout 400 python bench.py --workload simclr --no-cpu-baseline --steps 20 --warmup 6 > $O/bench_simclr.json 2> $O/bench_simclr.err
tail -4 $O/tests_gpu.log; tail -2 $O/smoke.log; head -c 600 $O/bench_moco.json; echo; head -c 300 $O/bench_moco2.json; echo; tail -3 $O/kernel_stats_serial.txt; tail -2 $O/pmc_traffic.txt
