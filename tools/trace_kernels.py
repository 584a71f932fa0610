#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace csv: launches, avg / min / max ms, workgroups.  usage: trace_kernels.py <dir> [skip_first_n_per_kernel]"""
import csv, glob, re, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = [r for f in files for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.OrderedDict()
for r in rows:
    if "xpng" not in r["Kernel_Name"]: continue
    k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("xpng::", "")
    k = re.sub(r"\(.*", "", k)
    key = (k, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    per.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(f"{'kernel':60s} {'wgs':>7s} {'thr':>5s} {'n':>4s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s}")
for (k, g, t), v in per.items():
    v = v[skip:] if len(v) > skip else v
    print(f"{k[:60]:60s} {g:7d} {t:5d} {len(v):4d} {sum(v)/len(v):9.3f} {min(v):9.3f} {max(v):9.3f}")
