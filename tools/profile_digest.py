#!/usr/bin/env python3
"""Digest of a rocprofv3 --kernel-trace csv of `bench.py`: per-kernel totals, and the roofline kernels' durations split into
the launches inside the pipelined steps (other kernels co-running) and the isolated launches of the roofline measurement
(the last `reps` launches of the batched grid).  usage: profile_digest.py <kernel_trace.csv> <bench.json> > digest.json"""
import csv, json, re, sys, collections
trace, benchj = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r): return re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "").replace("xpng::", "")
def wg(r): return int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
per = collections.defaultdict(list)
for r in rows:
    if "xpng" in r["Kernel_Name"]: per[(name(r), wg(r))].append(dur(r))
kern = [{"kernel": k, "workgroups": g, "launches": len(v), "total_ms": round(sum(v), 3), "avg_ms": round(sum(v) / len(v), 4),
         "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)} for (k, g), v in sorted(per.items(), key=lambda x: -sum(x[1]))]
# the roofline measurement = the longest run of consecutive chooser / transform launches on the batched grid with no other
# kernel of the library in between (the pipelined steps interleave other kernels)
xp = [r for r in rows if "xpng" in r["Kernel_Name"]]
# (the headline leg's pair: RGBA unless the trace holds none)
pair = ("k_chooser<4>", "k_m1_transform_rgba") if any(name(r).startswith("k_m1_transform_rgba") for r in xp) else ("k_chooser", "k_m1_transform")
def is_roof(r): return name(r).startswith(pair)
best, cur = [], []
for r in xp:
    if is_roof(r): cur.append(r)
    else:
        if len(cur) > len(best): best = cur
        cur = []
if len(cur) > len(best): best = cur
roof = {}
for key in pair:
    iso = [dur(r) for r in best if name(r).startswith(key)]
    if not iso: continue
    k, g = next((name(r), wg(r)) for r in best if name(r).startswith(key))
    allv = per[(k, g)]
    rest = len(allv) - len(iso)
    roof[k] = {"workgroups": g, "isolated_launches": len(iso), "isolated_avg_ms": round(sum(iso) / len(iso), 4),
               "in_pipeline_launches": rest, "in_pipeline_avg_ms": round((sum(allv) - sum(iso)) / max(1, rest), 4)}
tot = sum(x["isolated_avg_ms"] for x in roof.values())
print(json.dumps({"bench_line": benchj, "roofline_kernels": roof, "roofline_isolated_sum_ms": round(tot, 4),
                  "bench_ms_per_launch": benchj["roofline"]["ms_per_launch"], "kernels": kern}, indent=1))
