#!/usr/bin/env python3
"""Where do the serial-chain waves of the pipelined bench run, and does it matter?  Registers the wave probe (xpnghip_debug_probe,
csrc/common.hpp), runs P slots x B rasters of encode+decode steps, and for every probed wave (kernel id, workgroup, SE / CU / SIMD
/ XCC, start, end) counts the OTHER probed waves that shared its SIMD and its CU while it ran.  Prints, per chain kernel, wave
duration against that co-residency.   usage: wave_probe.py [P=4] [B=64] [steps=12]"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ["XPNG_USE_PROBES_LIB"] = "1"  # the switches this tool uses exist only in libxpng_hip_probes.so (make probes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import xpng_amd
from xpng_amd import api
from xpng_amd.synth import synth_raster_torch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
W = 4096
L = api.hip_lib()
L.xpnghip_debug_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
L.xpnghip_debug_probe_count.restype = ctypes.c_int64
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
rp = [r.data_ptr() for r in rs]
slots = []
for p in range(P):
    ctx = xpng_amd.Context(W, W, 4, batch=B)
    blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    slots.append(dict(ctx=ctx, bp=[t.data_ptr() for t in blobs], op=[t.data_ptr() for t in outs], stream=torch.cuda.Stream(), keep=(blobs, outs)))
lens = None
for sl in slots:
    lens = sl["ctx"].encode_device_batch(1, rp, sl["bp"])
    sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"])
torch.cuda.synchronize()
def step(k):
    sl = slots[k % P]; sh = sl["stream"].cuda_stream
    sl["ctx"].encode_device_batch(1, rp, sl["bp"], stream=sh, sync=False)
    sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"], stream=sh)
for k in range(2 * P): step(k)
torch.cuda.synchronize()
cap = 1 << 20
buf = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
assert L.xpnghip_debug_probe(buf.data_ptr(), cap) == 0
t0 = time.perf_counter()
for k in range(steps): step(k)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
n = int(L.xpnghip_debug_probe_count())
L.xpnghip_debug_probe(None, 0)
rec = np.frombuffer(buf.cpu().numpy().tobytes()[: min(n, cap) * 32], dtype=np.dtype([("k", "<u4"), ("blk", "<u4"), ("hw", "<u4"), ("xcc", "<u4"), ("t0", "<u8"), ("t1", "<u8")]))
print(f"P={P} B={B}: {ms:.2f} ms per step, {n} probed waves")
hw = rec["hw"]
wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xcc = rec["xcc"] & 15
cuid = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
simdid = cuid * 4 + simd
print("distinct CUs seen", len(np.unique(cuid)), "distinct SIMDs", len(np.unique(simdid)), "xcc values", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu))
dur = (rec["t1"] - rec["t0"]).astype(np.float64) / 100e3  # ms
has_iss = (rec["xcc"] & 16) != 0
issf = np.where(has_iss, (rec["xcc"] >> 16).astype(np.float64) / 65536.0, 0.0)   # fraction of its life in the boundary's landing / stores / request issue
cyc = np.where(has_iss, 0.0, (rec["xcc"] >> 5).astype(np.float64) * 16.0)        # shader-clock cycles of the wave (s_memtime)
long_ = (dur > 5) & ~has_iss
if long_.any():
    mhz = cyc[long_] / (dur[long_] * 1e3)
    print(f"shader clock over the waves longer than 5 ms: p10 {np.percentile(mhz,10):.0f} median {np.median(mhz):.0f} p90 {np.percentile(mhz,90):.0f} MHz")
names = {1: "chain2<ctx>", 2: "chain2<alpha>", 3: "dec_chain<ctx>", 4: "dec_chain<alpha>", 5: "walk_wide", 6: "walk(SALU)"}
# co-residency: time-weighted number of other probed waves on the same SIMD / CU
order = np.argsort(rec["t0"])
def coresid(ids):
    out = np.zeros(len(rec))
    by = {}
    for i in order: by.setdefault(int(ids[i]), []).append(i)
    for lst in by.values():
        for a in lst:
            ta0, ta1 = rec["t0"][a], rec["t1"][a]
            ov = 0
            for b in lst:
                if b == a: continue
                o = min(int(ta1), int(rec["t1"][b])) - max(int(ta0), int(rec["t0"][b]))
                if o > 0: ov += o
            out[a] = ov / max(1, int(ta1 - ta0))
    return out
cs, cc = coresid(simdid), coresid(cuid)
# launch instances: the i-th appearance of (kernel, workgroup) belongs to the kernel's i-th launch
inst = np.zeros(len(rec), dtype=np.int64)
seen = {}
for i in order:
    key = (int(rec["k"][i]), int(rec["blk"][i]) & 0xFFFF)
    inst[i] = seen.get(key, 0); seen[key] = inst[i] + 1
for k in sorted(names):
    m = rec["k"] == k
    if not m.any(): continue
    d = dur[m]
    # how late after its kernel's first wave does a wave start, and how long does the kernel take from first start to last end?
    delays, spans, longest = [], [], []
    for it in np.unique(inst[m]):
        q = m & (inst == it)
        if q.sum() < 2: continue
        st = rec["t0"][q].min()
        delays.append((rec["t0"][q] - st).astype(np.float64) / 100e3)
        spans.append((rec["t1"][q].max() - st) / 100e3); longest.append(dur[q].max())
    if delays:
        dl = np.concatenate(delays)
        print(f"{names[k]:18s} kernel span first start -> last end: median {np.median(spans):6.2f} ms; longest wave of a launch: median {np.median(longest):6.2f} ms; wave start delay after the launch's first wave: median {np.median(dl):6.2f}, p90 {np.percentile(dl,90):6.2f}, max {dl.max():6.2f} ms")
    wf = (rec["blk"][m] >> 16).astype(np.float64) / 65536.0
    if wf.max() > 0:
        lng = d >= np.median(d)
        print(f"{names[k]:18s} time standing at the block boundary's wait (long waves): median {np.median(wf[lng]) * 100:4.1f} %, p90 {np.percentile(wf[lng], 90) * 100:4.1f} % of the wave's life = {np.median(wf[lng] * d[lng]):5.2f} ms median")
    if has_iss[m].any():
        lng = d >= np.median(d); f = issf[m]
        print(f"{names[k]:18s} time in the rest of the boundary (landing, stores, issuing the next requests; long waves): median {np.median(f[lng]) * 100:4.1f} %, p90 {np.percentile(f[lng], 90) * 100:4.1f} % = {np.median(f[lng] * d[lng]):5.2f} ms median")
    print(f"{names[k]:18s} waves {m.sum():6d}  duration ms: p10 {np.percentile(d,10):7.2f} median {np.median(d):7.2f} p90 {np.percentile(d,90):7.2f} | mean other chain waves on its SIMD {cs[m].mean():5.2f}, on its CU {cc[m].mean():5.2f}")
    # does co-residency explain the duration?  long waves only (the kernel's size class): top half by duration
    big = m & (dur >= np.median(d))
    for lo, hi in ((0, 0.5), (0.5, 1.5), (1.5, 3), (3, 99)):
        q = big & (cs >= lo) & (cs < hi)
        if q.sum() > 5: print(f"      other chain waves on the SIMD in [{lo}, {hi}): {q.sum():5d} waves, median duration {np.median(dur[q]):7.2f} ms")
    for lo, hi in ((0, 3), (3, 6), (6, 9), (9, 12), (12, 16), (16, 99)):
        q = big & (cc >= lo) & (cc < hi)
        if q.sum() > 5: print(f"      other chain waves on the CU   in [{lo}, {hi}): {q.sum():5d} waves, median duration {np.median(dur[q]):7.2f} ms")
    # where: per XCC, and by workgroup index (the launch's dispatch order)
    if m.sum() >= 512 and np.median(d) > 5:
        print("      by XCC: " + "  ".join(f"{x}: {np.median(dur[m & (xcc == x)]):5.1f}" for x in np.unique(xcc[m])))
        blk = rec["blk"][m] & 0xFFFF; nb = int(blk.max()) + 1
        qs = [(i * nb // 4, (i + 1) * nb // 4) for i in range(4)]
        print("      by workgroup index quarter: " + "  ".join(f"[{a},{b}): {np.median(d[(blk >= a) & (blk < b)]):5.1f}" for a, b in qs))
        percu = {}
        for c_, dd in zip(cuid[m], d): percu.setdefault(int(c_), []).append(dd)
        meds = np.array([np.median(v) for v in percu.values() if len(v) >= 3])
        if len(meds) > 8: print(f"      per-CU median duration over {len(meds)} CUs: p10 {np.percentile(meds,10):5.1f} median {np.median(meds):5.1f} p90 {np.percentile(meds,90):5.1f}")
