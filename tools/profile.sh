#!/bin/bash
# rocprofv3 profiles of the bench command (run on the GPU box via gpurun).  Counters are collected in
# their own passes (one pass per counter group), never together with sys/runtime tracing.
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
ARGS=${2:---steps 5 --warmup 2 --no-cpu-baseline}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.log" 2>&1
find "$OUT" -name '*.csv' | head -50
# keep the merge-back small: the per-dispatch traces can be large
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
