#!/usr/bin/env python3
"""Does the transform find the rows the chooser has just read in the Infinity Cache?  Chooser + transform over 64 distinct 4096^2 RGBA
rasters as ONE launch pair (bench.py's roofline measurement) against the same work in groups of G rasters (chooser(group) then
transform(group): the group's sampled rows, half of each raster, are at most G x 32 MB old when the transform wants them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch
B, W = 64, 4096
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
rp = [r.data_ptr() for r in rs]
ctx = xpng_amd.Context(W, W, 4, batch=B)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for G in (64, 16, 8, 4, 2, 1):
    def run():
        for k in range(0, B, G): ctx.transform_device_batch(rp[k:k + G], stream=st)
    ms = timed(run)
    print(f"groups of {G:2d}: {ms:6.3f} ms per 64 rasters  -> {10 * B * W * W / ms / 1e6 / 8000:.3f} of HBM peak on 10 B/px", flush=True)
