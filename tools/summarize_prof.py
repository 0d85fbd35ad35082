#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into profiles/: per-kernel time (rocprofv3 --stats)
and per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read, so the read side is
doubled ("fetch x2"); WRITE_SIZE is exact for 16-B streaming stores.  Other access widths are
uncalibrated, so the raw number is kept beside the corrected one.

usage: summarize_prof.py <profile dir> <round tag> [metric]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PHASE_OF = {
    "k_bounds": "bounds", "k_bounds_final": "bounds",
    # grid build by partition (round 2) / by sort (fallback for huge grids)
    "k_part_hist": "part_hist", "k_part_scatter": "part_scatter", "k_part_fine": "part_fine",
    "k_cell_key": "cell_key", "rocprim_radix_sort": "cell_sort", "k_tile_marks": "cell_scan",
    "k_cellstart_tiles": "cell_scan", "k_gather": "scatter",
    # the small scans (chunk counts, work lists, seed ranks) are spread over three phases: listed, not attributed
    "k_core": "core_count", "k_core_lds": "core_count", "k_wl_fill": "core_count",
    "k_union": "union", "k_union_init": "union", "k_union_init_list": "union", "k_flatten0": "union",
    "k_flatten": "flatten_number", "k_seed_popc": "flatten_number", "k_seedflag": "flatten_number",
    "k_rootk": "flatten_number",
    "k_labk_rest": "border", "k_border": "border", "k_border_list": "border",
    "k_output": "output", "k_out_scatter": "out_scatter", "k_out_write": "out_write",
    "k_icp_pass": "icp", "k_icp_pass_small": "icp", "k_icp_step": "icp", "k_model32": "icp", "k_absmax": "icp",
}


def short(name):
    if "rocprim" in name and ("radix" in name or "onesweep" in name or "histogram" in name):
        return "rocprim_radix_sort"
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def main():
    src = sys.argv[1]
    tag = sys.argv[2]
    metric = sys.argv[3] if len(sys.argv) > 3 else "L1_2D"
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = []
    def newest(pattern):
        # gpurun merges outputs into the local directory without deleting older runs: take the latest file
        fs = sorted(glob.glob(pattern), key=os.path.getmtime)
        return fs[-1:]

    for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
    pmc = {}
    for key, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        for f in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] == key:
                        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[key] = {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}  # KiB -> bytes per launch
    lines = ["# rocprofv3 summary %s (%s)" % (tag, metric), "",
             "Source: `tools/profile.sh` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 "
             "--warmup 2 --no-cpu-baseline --no-extras`, plus one `--pmc FETCH_SIZE` and one `--pmc WRITE_SIZE` pass.",
             "FETCH/WRITE are average bytes per launch; `fetch x2` applies the gfx950 correction for wide "
             "coalesced reads (uncalibrated for narrow/gather accesses, so both are shown).", "",
             "| kernel | calls | avg us | % | FETCH_SIZE B | fetch x2 B | WRITE_SIZE B |", "|---|---|---|---|---|---|---|"]
    per_phase = defaultdict(lambda: {"fetch_raw": 0.0, "write": 0.0, "avg_us": 0.0})
    # template instantiations of one kernel are one line: calls and time summed, average = total / calls
    agg, order = {}, []
    for r in rows:
        k = short(r["Name"])
        if k not in agg:
            agg[k] = [0, 0.0, 0.0]
            order.append(k)
        agg[k][0] += int(r["Calls"])
        agg[k][1] += float(r["TotalDurationNs"])
        agg[k][2] += float(r["Percentage"])
    # launches of a once-per-step kernel = profiled steps
    steps = max(agg.get("k_out_write", agg.get("k_output", [1]))[0], 1)
    for k in order:
        calls, tot, pct = agg[k]
        fz = pmc["FETCH_SIZE"].get(k)
        wz = pmc["WRITE_SIZE"].get(k)
        lines.append("| %s | %d | %.1f | %.2f | %s | %s | %s |" % (
            k, calls, tot / calls / 1e3, pct,
            "%.3e" % fz if fz is not None else "-", "%.3e" % (2 * fz) if fz is not None else "-",
            "%.3e" % wz if wz is not None else "-"))
        ph = PHASE_OF.get(k)
        if ph and fz is not None and wz is not None:
            per_step = calls / steps  # a kernel launched several times per step (the scans) counts that often
            per_phase[ph]["fetch_raw"] += fz * per_step
            per_phase[ph]["write"] += wz * per_step
            per_phase[ph]["avg_us"] += tot / steps / 1e3
    extras = newest(os.path.join(src, "stats_extras", "*", "*_kernel_stats.csv"))
    if extras:
        with open(extras[0]) as fh:
            erows = list(csv.DictReader(fh))
        lines += ["", "Kernels of the side measurements (`bench.py` with its extras: ICP 1 M x 100, centroids / ICP / matching "
                  "after the clustering, block pipeline), kernel time only:", "",
                  "| kernel | calls | avg us | total ms |", "|---|---|---|---|"]
        eagg, eorder = {}, []
        for r in erows:
            k = short(r["Name"])
            if k in agg and not k.startswith("k_icp"):
                continue  # already in the table above
            if k not in eagg:
                eagg[k] = [0, 0.0]
                eorder.append(k)
            eagg[k][0] += int(r["Calls"])
            eagg[k][1] += float(r["TotalDurationNs"])
        for k in eorder[:30]:
            lines.append("| %s | %d | %.1f | %.2f |" % (k, eagg[k][0], eagg[k][1] / eagg[k][0] / 1e3, eagg[k][1] / 1e6))
    with open(os.path.join(out_dir, "%s_%s_rocprof.md" % (tag, metric)), "w") as f:
        f.write("\n".join(lines) + "\n")
    latest = os.path.join(out_dir, "pmc_latest.json")
    data = {}
    if os.path.exists(latest):
        with open(latest) as f:
            data = json.load(f)
    data[metric] = {ph: v["fetch_raw"] * 2 + v["write"] for ph, v in per_phase.items()}
    data[metric + "_detail"] = {ph: v for ph, v in per_phase.items()}
    data["_note"] = ("bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB->B), summed over the kernels of a phase; "
                     "round tag " + tag)
    with open(latest, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
