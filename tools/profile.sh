#!/bin/bash
# rocprofv3 profiles of the bench command (run on the GPU box via gpurun).  Counters are collected in
# their own passes (one pass per counter group), never together with sys/runtime tracing.
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
ARGS=${2:---steps 5 --warmup 2 --no-cpu-baseline --no-extras}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.log" 2>&1
# kernel times of the side measurements (ICP, block pipeline, centroids): stats only, L1_2D default run
if [ -z "$2" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_extras" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_extras.log" 2>&1
fi
find "$OUT" -name '*.csv' | head -50
# keep the merge-back small: the per-dispatch traces can be large
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
