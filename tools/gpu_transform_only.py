#!/usr/bin/env python3
"""Runs only the BASELINE config-2 kernels (k_chooser + k_m1_transform) for rocprofv3 --pmc / --stats passes.
usage: gpu_transform_only.py [image_edge=4096] [batch=16] [reps=5] [rgb]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
alpha = not (len(sys.argv) > 4 and sys.argv[4] == "rgb")
# B DISTINCT rasters (different seeds) so that reads really come from HBM, not from the 256 MB Infinity Cache
rs = [synth_raster_torch("photo", W, W, alpha, seed=1 + b) for b in range(B)]
ctx = xpng_amd.Context(W, W, 4 if alpha else 3, batch=B)
ptrs = [r.data_ptr() for r in rs]
for _ in range(reps):
    ctx.transform_device_batch(ptrs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s = torch.cuda.current_stream().cuda_stream
e0.record()
for _ in range(reps):
    ctx.transform_device_batch(ptrs, stream=s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
px = W * W * B
print(f"{W}x{W} x{B} {'RGBA' if alpha else 'RGB'}: {ms*1e3:.1f} us/launch  {px/ms/1e6:.1f} Gpx/s  algorithmic {(10.0 if alpha else 7.75)*px/ms/1e9:.2f} TB/s")
