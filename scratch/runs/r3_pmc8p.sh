#!/bin/bash
# PMC passes over the 8-phase kernel (staged / persistent) and the ring kernel on two shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3_pmc8p; rm -rf $O; mkdir -p $O
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
cd /tmp
i=0
for shape in "50432 768 2304" "8192 8192 8192"; do
 for v in "0 0" "2 0" "2 1"; do
  i=$((i+1))
  C="python $GRAFT_REPO_ROOT/scratch/pmc_8p.py $shape $v 4"
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pa$i -o a -- $C > $O/log_a$i.txt 2>&1
  echo "== shape $shape  mode/direct $v" >> $O/stall.txt
  python $GRAFT_REPO_ROOT/tools/pmc_stall_summary.py $(db /tmp/pa$i) >> $O/stall.txt 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/pb$i -o b -- $C > $O/log_b$i.txt 2>&1
  echo "== shape $shape  mode/direct $v" >> $O/busy.txt
  python $GRAFT_REPO_ROOT/tools/pmc_stall_summary.py $(db /tmp/pb$i) >> $O/busy.txt 2>&1
 done
done
cat $O/stall.txt; cat $O/busy.txt
