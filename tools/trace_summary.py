#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace csv, optionally only dispatches with >= MIN workgroups (batched steps)."""
import csv, sys, re, collections
path = sys.argv[1]; minwg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    if wg < minwg: continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "")
    a = acc[(k, wg)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (k, wg), (n, t) in sorted(acc.items(), key=lambda x: -x[1][1])[:30]:
    print(f"{k[:50]:50s} wg={wg:7d} n={n:4d} total={t:9.2f} ms avg={t/n:8.3f} ms")
