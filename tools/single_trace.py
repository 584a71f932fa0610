#!/usr/bin/env python3
"""ONE image, strictly serial encode / decode launches (for rocprofv3 --kernel-trace: which kernel owns the single-image latency).
usage: single_trace.py rgba|rgb [level=1] [reps=6] [size=4096] [batch=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch
alpha = sys.argv[1] == "rgba"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
W = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ch = 4 if alpha else 3
rs = [synth_raster_torch("photo", W, W, alpha, seed=1 + b) for b in range(B)]
ctx = xpng_amd.Context(W, W, ch, batch=B)
blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
outs = [torch.zeros(W * W * ch, dtype=torch.uint8, device="cuda") for _ in range(B)]
rp, bp, op = [r.data_ptr() for r in rs], [b.data_ptr() for b in blobs], [o.data_ptr() for o in outs]
lens = ctx.encode_device_batch(level, rp, bp)
ctx.decode_device_batch(level, bp, lens, None, op)
torch.cuda.synchronize()
assert ctx.decode_status() == 0 and all(torch.equal(outs[b].view_as(rs[b]), rs[b]) for b in range(B))
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e = timed(lambda: ctx.encode_device_batch(level, rp, bp, sync=False))
d = timed(lambda: ctx.decode_device_batch(level, bp, lens, None, op))
print(f"{sys.argv[1]} level {level} {W}x{W} B={B}: encode {e:.2f} ms, decode {d:.2f} ms (wall, one launch sequence at a time)")
