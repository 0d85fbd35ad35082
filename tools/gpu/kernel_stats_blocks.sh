#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/profb
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_blocks.py > $OUT/log.txt 2>&1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:40]:
    nm=r["Name"]
    m=re.search(r"(k_[a-z0-9_]+)", nm)
    short = m.group(1) if m else ("rocprim:"+re.sub(r".*detail::([a-z_]+).*", r"\1", nm)[:30] if "rocprim" in nm else nm[:40])
    per_iter=float(r["TotalDurationNs"])/4/1e3
    print("%-40s calls/iter %6.1f avg %8.1f us  per-iter %8.1f us" % (short, int(r["Calls"])/4, float(r["AverageNs"])/1e3, per_iter))
PY
tail -5 $OUT/log.txt
find $OUT -name '*kernel_trace.csv' -delete
