#!/usr/bin/env python3
"""Level-2 (RGB) batch encode / decode, strictly serial launches (for rocprofv3 --kernel-trace).  usage: l2_trace.py [B=32] [reps=3] [level=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
level = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W = 4096
rs = [synth_raster_torch("photo", W, W, False, seed=1 + b) for b in range(B)]
ctx = xpng_amd.Context(W, W, 3, batch=B)
blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
outs = [torch.zeros(W * W * 3, dtype=torch.uint8, device="cuda") for _ in range(B)]
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
print(f"RGB level {level} B={B}: encode {e:.2f} ms ({B*W*W/e/1e6:.1f} Gpx/s), decode {d:.2f} ms ({B*W*W/d/1e6:.1f} Gpx/s)")
