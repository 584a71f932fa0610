#!/usr/bin/env python3
"""Digest of tools/prof_pmc_transform.sh passes -> profiles/rNN_pmc_transform_<mode>.json (what bench.py's roofline.traffic reads).
usage: pmc_transform_digest.py <gpurun_out dir> <mode: rgba|rgb> <batch>"""
import csv, glob, json, re, sys, collections
root, mode, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
px = 4096 * 4096 * B
def load(tag):
    acc, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    seen = set()
    for f in glob.glob(f"{root}/pmct_{mode}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "").replace("xpng::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return {k: {c: v / n[k] for c, v in cs.items()} for k, cs in acc.items()}
a, d, e = load("a"), load("d"), load("e")
kern, fetch, write = {}, 0.0, 0.0
pp = {}
for k in sorted(set(d) | set(e)):
    f = d.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2   # KiB; doubled: gfx950 reports half the bytes of wide coalesced reads
    w = e.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    fetch += f; write += w
    short = "chooser" if "chooser" in k else "transform"
    pp[short + "_fetch"] = f / px; pp[short + "_write"] = w / px
    kern[k] = {"fetch_bytes_corrected": int(f), "write_bytes": int(w), **{c: round(v) for c, v in a.get(k, {}).items()}}
pp["total"] = (fetch + write) / px
algo = (10.0 if mode == "rgba" else 7.75) * px
out = {"command": f"tools/prof_pmc_transform.sh {B} {mode}  =  rocprofv3 --kernel-include-regex xpng --pmc <set> -- python3 tools/gpu_transform_only.py 4096 {B} 3 {mode} (one pass per counter set, counters only; {B} DISTINCT 4096^2 {mode.upper()} rasters per launch)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch as reported; FETCH_SIZE doubled for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)",
       "images_per_launch": B, "per_launch_bytes": {"fetch_corrected": int(fetch), "write": int(write), "total": int(fetch + write), "algorithmic": int(algo)},
       "per_pixel_bytes": pp,
       "valu_lane_instructions_per_pixel": {("chooser" if "chooser" in k else "transform"): v.get("SQ_INSTS_VALU", 0) * 64 / px for k, v in a.items()},
       "kernels": kern}
print(json.dumps(out, indent=1))
