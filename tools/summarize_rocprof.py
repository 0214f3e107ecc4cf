#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC counter collections) into one small per-kernel table.

  python tools/summarize_rocprof.py --stats gpurun_out/kt --pmc gpurun_out/pmc_f gpurun_out/pmc_w --pairs-per-launch 16 -o profiles/x.csv
"""
import argparse
import collections
import csv
import glob
import os
import re


def short(name):
    name = name.split("(")[0]
    if name.startswith("void at::native"):
        return "torch::" + name.split("::")[2].split("<")[0]
    name = name.replace("sv::", "")
    if name.startswith("void "):  # template instantiations: "void k_dense<false>" -> "k_dense" (the <true> variants are the counting builds)
        name = name[5:]
    if "<false" in name:  # "k_dense<false, 4, 3>" (not counting, 4 mask words, plane radius 3) -> "k_dense"
        name = name[:name.index("<false")]
    name = re.sub(r"<\d+>$", "", name)  # "k_ccl_band<256>" / "<1024>" (workgroup size of the launch) -> "k_ccl_band"
    return name


ap = argparse.ArgumentParser()
ap.add_argument("--stats", help="directory of a --kernel-trace --stats run")
ap.add_argument("--pmc", nargs="*", default=[], help="directories of --pmc runs")
ap.add_argument("--pairs-per-launch", type=float, default=16)
ap.add_argument("-o", "--out", required=True)
a = ap.parse_args()

rows = collections.OrderedDict()
if a.stats:
    f = max(glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)  # the newest run in the directory
    # medians from the per-dispatch trace of the same run (the stats file has average / min / max only; a cold first launch skews the average)
    durs = collections.defaultdict(list)
    tr = glob.glob(os.path.join(os.path.dirname(f), "*kernel_trace.csv"))
    if tr:
        for r in csv.DictReader(open(max(tr, key=os.path.getmtime))):
            durs[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        d = sorted(durs.get(k, []))
        med = d[len(d) // 2] if len(d) % 2 else (0.5 * (d[len(d) // 2 - 1] + d[len(d) // 2]) if d else float(r["AverageNs"]) / 1e3)
        if k in rows:  # several instantiations of one kernel (k_dense<..>): keep the one with the most time
            if float(r["TotalDurationNs"]) <= rows[k]["_total"]:
                continue
        rows[k] = {"kernel": k, "calls": int(r["Calls"]), "median_us": round(med, 2), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "min_us": round(float(r["MinNs"]) / 1e3, 2),
                   "max_us": round(float(r["MaxNs"]) / 1e3, 2), "pct": r["Percentage"], "us_per_pair": round(med / a.pairs_per_launch, 3), "_total": float(r["TotalDurationNs"])}
    for r in rows.values():
        r.pop("_total", None)
for d in a.pmc:
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    for k, cs in acc.items():
        row = rows.setdefault(k, {"kernel": k})
        for c, v in cs.items():
            row[c + "_per_launch"] = round(v / n[k][c], 1)
cols = []
for r in rows.values():
    for c in r:
        if c not in cols:
            cols.append(c)
with open(a.out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=cols)
    w.writeheader()
    for r in rows.values():
        w.writerow(r)
print("wrote", a.out, len(rows), "kernels")
