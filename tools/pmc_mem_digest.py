#!/usr/bin/env python3
"""Per-kernel HBM bytes of one batched step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_mem_digest.py <fetch_dir> <write_dir> <batch> [bench flags...]  -> JSON on stdout
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB per dispatch; FETCH_SIZE reports half the bytes of wide
coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both are calibrated for streaming
access only: for the byte-granular kernels of the entropy stage the figures are indicative."""
import csv, glob, json, re, sys, collections
fdir, wdir, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
rgb = "--rgb" in sys.argv
def load(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "xpng" not in r["Kernel_Name"]: continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xpng::", "").replace("xpng::", "")
            key = (k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            acc[key] += float(r["Counter_Value"]); n[key] += 1
    return {k: acc[k] / n[k] for k in acc}, n
fe, nf = load(fdir, "FETCH_SIZE")
wr, nw = load(wdir, "WRITE_SIZE")
px = 4096 * 4096 * B
# the batched launches: for every kernel name keep the dispatch group with the most workgroups (single-image launches of the
# verification / latency legs use the same kernels on small grids)
best = {}
for (k, g) in set(fe) | set(wr):
    if k not in best or g > best[k]: best[k] = g
# kernels (or template instances) that only the single-image launches of the bench's verification / latency legs use: the narrow
# entropy kernels, the 1024-thread walkers, the anti-diagonal reconstruction, xpng_store's normalisation.  They are not part of the
# batched step (rounds 2-3 counted them: ~0.6 B/px of the 47.5 of profiles/r03_final_pmc_step_mem.json)
SINGLE = re.compile(r"^k_rans2_encode$|^k_rans2_decode<|^k_dec_recon<|, 1024|<1024>|^k_norm_|^k_any_differs|^k_rans1_encode$|^k_rans1_decode")
rows, tf, tw, sf, sw = [], 0.0, 0.0, 0.0, 0.0
for (k, g) in sorted(set(fe) | set(wr)):
    if g * 7 < best[k]: continue         # a single-image launch of the same kernel (the decode tail splits its launches by
    f = fe.get((k, g), 0.0) * 1024 * 2   # tile size class: both parts of such a split are batched launches and count)
    w = wr.get((k, g), 0.0) * 1024       # KiB -> bytes; fetch x2 (gfx950 wide-read correction)
    single = bool(SINGLE.search(k))
    if single: sf += f; sw += w
    else: tf += f; tw += w
    rows.append({"kernel": k, "workgroups": g, "fetch_bytes_corrected": int(f), "write_bytes": int(w), "bytes_per_px": round((f + w) / px, 3), **({"single_image_launch": True} if single else {})})
rows.sort(key=lambda r: -(r["fetch_bytes_corrected"] + r["write_bytes"]))
pxsz = 3 if rgb else 4
algo = px * (pxsz + 1.2155) * 2   # encode: read PXSZ + write compressed; decode: read compressed + write PXSZ
print(json.dumps({"what": f"one encode+decode step over {B} distinct 4096^2 {'RGB' if rgb else 'RGBA'} rasters, one pipeline slot; per-dispatch averages of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)",
                  "pixels_per_step": px, "fetch_bytes_corrected": int(tf), "write_bytes": int(tw), "total_bytes": int(tf + tw),
                  "bytes_per_px": round((tf + tw) / px, 2), "bytes_per_px_with_single_image_launches": round((tf + tw + sf + sw) / px, 2), "algorithmic_bytes": int(algo), "algorithmic_bytes_per_px": round(algo / px, 2),
                  "traffic_over_algorithmic": round((tf + tw) / algo, 2), "kernels": rows}, indent=1))
