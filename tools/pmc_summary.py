#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files (several passes merged by kernel name)."""
import csv, sys, re, json, collections
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]; grid = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        key = (k, grid)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (key, r["Dispatch_Id"])
        if d not in seen: seen.add(d); n[key] += 1
    return acc, n
out = {}
for p in sys.argv[1:]:
    acc, n = load(p)
    for key, cs in acc.items():
        e = out.setdefault(key, {"dispatches": n[key]})
        for c, v in cs.items(): e[c] = v / n[key]   # per-dispatch average
rows = []
for (k, grid), e in out.items():
    if "xpng" not in k: continue
    nm = re.sub(r"\(.*", "", k); nm = nm.replace("void xpng::", "").replace("xpng::", "")
    rows.append((nm, grid, e))
rows.sort(key=lambda r: -r[2].get("SQ_INSTS_VALU", 0))
for nm, grid, e in rows:
    print(json.dumps({"kernel": nm, "workgroups": grid, **{k: (round(v) if isinstance(v, float) else v) for k, v in e.items()}}))
