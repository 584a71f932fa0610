#!/usr/bin/env python3
"""Short per-kernel table out of a rocprofv3 --stats kernel_stats.csv.  usage: kstats.py <dir-or-csv> [rows]"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
k = 0
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void xpng::", "").replace("xpng::", "")
    if "at::" in n or "rocclr" in n:
        continue
    print("%-46s calls %5s avg %8.3f min %8.3f max %8.3f  %s%%" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
    k += 1
    if k >= n_rows:
        break
