#!/usr/bin/env python3
"""Summarise tools/gpu/sq_counters.sh's two --pmc passes (gpurun_out/pmc3) into profiles/<tag>_sq_counters.md:
per kernel the average per launch of each counter, summed over the XCDs.  usage: summarize_sq.py <dir> <tag>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
files = []
for d in sorted(glob.glob(os.path.join(src, "p*"))):  # gpurun merges runs without deleting older ones: newest per pass
    fs = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    files += fs[-1:]
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if not m:
                continue
            acc[m.group(1)][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
cols = ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD",
        "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU"]
rows = []
for k, cs in acc.items():
    avg = {c: (sum(v.values()) / len(v) if v else 0.0) for c, v in cs.items()}
    rows.append((avg.get("GRBM_GUI_ACTIVE", 0.0), k, avg))
rows.sort(reverse=True)
out = ["# SQ counters %s (MI355X, C4 10 M points, L1_2D)" % tag, "",
       "Source: `tools/gpu/sq_counters.sh` = two `rocprofv3 --pmc ... --kernel-trace` passes of `python3 bench.py --steps 3 "
       "--warmup 1 --no-cpu-baseline --no-extras`; averages per launch, summed over the 8 XCDs "
       "(`tools/summarize_sq.py`).", "",
       "| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols[:5]) + " | lanes / VALU inst | "
       + " | ".join(c.replace("SQ_", "") for c in cols[5:]) + " |",
       "|---|" + "---|" * (len(cols) + 1)]
for _, k, a in rows[:20]:
    lanes = a.get("SQ_THREAD_CYCLES_VALU", 0.0) / a["SQ_INSTS_VALU"] if a.get("SQ_INSTS_VALU") else 0.0
    cells = ["%.3g" % a.get(c, 0.0) for c in cols[:5]] + ["%.1f" % lanes] + ["%.3g" % a.get(c, 0.0) for c in cols[5:]]
    out.append("| %s | %s |" % (k, " | ".join(cells)))
out += ["", "`lanes / VALU inst` = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (64 = no divergence)."]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_sq_counters.md" % tag)
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[4:14]))
