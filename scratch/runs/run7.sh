#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_run7
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $GRAFT_REPO_ROOT/$O/pmc1 -o p -- python $GRAFT_REPO_ROOT/scratch/pmc_convs.py > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES -d $GRAFT_REPO_ROOT/$O/pmc2 -o p -- python $GRAFT_REPO_ROOT/scratch/pmc_convs.py > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 4 $O/ops.log; ls -la $O/pmc1 $O/pmc2
