#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY TCP_READ_TAGCONFLICT_STALL_CYCLES TA_FLAT_READ_WAVEFRONTS TCP_GATE_EN1" \
           "TCC_HIT TCC_MISS TCC_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TOTAL_READ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/log$i.txt 2>&1 || { tail -5 $OUT/log$i.txt; echo "pass $i failed"; }
done
python3 - <<'PY'
import csv,glob,re,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc2/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=sorted({c for k in acc for c in acc[k]})
import json
out={k:{c:sum(v)/len(v) for c,v in acc[k].items()} for k in acc}
json.dump(out,open("gpurun_out/pmc2/summary.json","w"),indent=1)
for k in ("k_core_lds","k_union_init","k_union","k_border","k_cell_hist","k_scatter"):
    if k in out:
        print(k, {c: ("%.3g"%v) for c,v in sorted(out[k].items())})
PY
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +5M -delete
