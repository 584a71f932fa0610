#!/bin/bash
# L2 (TCC) and L1 (TCP) request counts of EVERY kernel of one batched encode+decode step: rocprofv3 counters-only passes over bench.py
# with one pipeline slot and one timed step (kernels are serialised under --pmc: per-kernel totals, not co-run behaviour).
# Output: gpurun_out/pmcl2_<n>/, digest (per kernel, largest grid = the batched launch) on stdout.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; shift || true
n=0
for set in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$((n+1))
  timeout -k 10 400 rocprofv3 --kernel-include-regex "xpng" --pmc $set --output-format csv -d $R/gpurun_out/pmcl2_$n -o pmc -- \
    python3 $R/bench.py --no-cpu --no-legs --no-config4 --batch $B --pipeline 1 --steps 1 --warmup 1 --roofline-reps 1 "$@" > $R/gpurun_out/pmcl2_$n.log 2>&1
  echo "pass $n ($set) rc $?"
done
python3 - $R/gpurun_out $B <<'PY'
import csv, glob, re, sys, collections, json
root, B = sys.argv[1], int(sys.argv[2])
acc, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter)
for f in glob.glob(root + "/pmcl2_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xpng" not in r["Kernel_Name"]: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "").replace("xpng::", "")
        key = (k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
best = {}
for (k, g) in acc:
    if k not in best or g > best[k]: best[k] = g
rows = []
for (k, g), c in acc.items():
    if g * 7 < best[k]: continue
    rows.append({"kernel": k, "workgroups": g, **{n: round(v / cnt[(k, g)][n] / 1e6, 2) for n, v in sorted(c.items())}})
rows.sort(key=lambda r: -r.get("TCC_REQ_sum", 0))
tot = collections.Counter()
for r in rows:
    for n, v in r.items():
        if n not in ("kernel", "workgroups"): tot[n] += v
print(json.dumps({"what": f"millions of requests per dispatch, one encode+decode step over {B} rasters (batched launches)", "total": {n: round(v, 1) for n, v in tot.items()}, "kernels": rows}, indent=1))
PY
