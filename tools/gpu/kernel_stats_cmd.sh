#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script: tools/gpu/kernel_stats_cmd.sh tools/bench_icp_kk.py
set -e
export TMPDIR=/tmp
OUT=gpurun_out/profc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/log.txt 2>&1
grep -v "^[EW]2026" $OUT/log.txt | tail -5
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    nm=r["Name"]; m=re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", nm)
    print("%-34s calls %6s avg %10.1f us total %10.1f ms" % ((m.group(0) if m else nm[:34])[:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $OUT -name '*kernel_trace.csv' -delete
