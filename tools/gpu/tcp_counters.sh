#!/bin/bash
# memory-pipeline counters, two per pass (larger sets are rejected by the profiler on gfx950)
export TMPDIR=/tmp
OUT=gpurun_out/pmc4
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "TCC_HIT TCC_MISS" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY" "TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_READ"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/log$i.txt 2>&1
  echo "pass $i ($set) rc=$?"
done
find $OUT -name '*kernel_trace.csv' -delete
exit 0
