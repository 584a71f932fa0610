#!/usr/bin/env python3
"""Timeline of the pipelined steps out of a rocprofv3 --kernel-trace csv: the batched launches (>= MINWG workgroups or chain
kernels) of the last SPAN ms, one line per launch, grouped by HIP stream.  usage: timeline.py <kernel_trace.csv> [span_ms] [stream-filter]"""
import csv, re, sys, collections
path = sys.argv[1]; span = float(sys.argv[2]) if len(sys.argv) > 2 else 130.0
rows = [r for r in csv.DictReader(open(path)) if "xpng::" in r["Kernel_Name"]]
def name(r): return re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "").replace("xpng::", "")
def wg(r): return int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
# pipelined steps = the dispatches of the batched grid; drop the isolated roofline / single-image launches at the start
big = [r for r in rows if wg(r) >= 60 and not name(r).startswith(("k_rans2_encode", "k_rans2_decode<", "k_dec_recon<", "k_m1_streams<4, 1024", "k_dec_resid<4, 1024", "k_dec_alpha<1024"))]
t_end = max(int(r["End_Timestamp"]) for r in big)
sel = [r for r in big if int(r["Start_Timestamp"]) >= t_end - span * 1e6]
t0 = min(int(r["Start_Timestamp"]) for r in sel)
by = collections.defaultdict(list)
for r in sel: by[r["Stream_Id"]].append(r)
for sid in sorted(by, key=lambda s: int(s)):
    print(f"--- stream {sid}")
    for r in sorted(by[sid], key=lambda r: int(r["Start_Timestamp"])):
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        print(f"  {a:8.2f} -> {b:8.2f}  ({b - a:7.2f} ms)  {name(r)[:44]:44s} wg={wg(r)}")
