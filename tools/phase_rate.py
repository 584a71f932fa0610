#!/usr/bin/env python3
"""Pipelined rate of ONE direction (encode or decode), P contexts x B distinct rasters: where does the combined rate come from?
usage: phase_rate.py enc|dec [B=64] [P=4] [steps=24]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.api import walk_tile_offsets
from xpng_amd.synth import synth_raster_torch
mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 24
W = 4096
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
slots = []
for p in range(P):
    ctx = xpng_amd.Context(W, W, 4, batch=B)
    blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)] if mode == "dec" else None
    slots.append(dict(ctx=ctx, blobs=blobs, outs=outs, stream=torch.cuda.Stream()))
lens = slots[0]["ctx"].encode_device_batch(1, [r.data_ptr() for r in rs], [b.data_ptr() for b in slots[0]["blobs"]])
offs = [walk_tile_offsets(slots[0]["blobs"][i][:lens[i]].cpu().numpy().tobytes(), slots[0]["ctx"].n_tiles)[0] for i in range(B)]
for sl in slots[1:]:
    sl["ctx"].encode_device_batch(1, [r.data_ptr() for r in rs], [b.data_ptr() for b in sl["blobs"]])
torch.cuda.synchronize()
def step(k):
    sl = slots[k % P]
    sh = sl["stream"].cuda_stream
    if mode == "enc":
        sl["ctx"].encode_device_batch(1, [r.data_ptr() for r in rs], [b.data_ptr() for b in sl["blobs"]], stream=sh, sync=False)
    else:
        sl["ctx"].decode_device_batch(1, [b.data_ptr() for b in sl["blobs"]], lens, offs, [o.data_ptr() for o in sl["outs"]], stream=sh)
for k in range(2 * P): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps): step(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{mode} B={B} P={P}: {dt / steps * 1e3:.2f} ms per {B} images = {B * W * W * steps / dt / 1e9:.2f} Gpx/s")
