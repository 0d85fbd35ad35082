#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/profq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $OUT/log.txt 2>&1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    m=re.search(r"(k_[a-z0-9_]+)", r["Name"]); print("%-22s calls %4s avg %9.1f us  %5s%%" % (m.group(1) if m else r["Name"][:22], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $OUT -name '*kernel_trace.csv' -delete
