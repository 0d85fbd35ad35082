#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/log$i.txt 2>&1
  echo "pass $i done"
done
find $OUT -name '*kernel_trace.csv' -delete
