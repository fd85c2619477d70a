#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lab
mkdir -p $OUT
for v in 96_1_2 96_3_2; do
  timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a_$v -- $R/scratch/labbin/lab_$v 42943 452 > $OUT/a_$v.log 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b_$v -- $R/scratch/labbin/lab_$v 42943 452 > $OUT/b_$v.log 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c_$v -- $R/scratch/labbin/lab_$v 42943 452 > $OUT/c_$v.log 2>&1 || exit 1
done
echo ok
