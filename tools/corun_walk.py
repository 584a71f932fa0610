#!/usr/bin/env python3
"""What slows the decode chains beside other work?  One context decodes a 64-image batch (serially, one batch at a time) while a
second context keeps ONE kind of other work in flight on its own stream:  none | transform | encode | decode.
Run under rocprofv3 --kernel-trace and compare the walk / chain durations (tools/trace_summary.py).
usage: corun_walk.py none|transform|encode|decode [reps=6]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.api import walk_tile_offsets
from xpng_amd.synth import synth_raster_torch
other = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, W = 64, 4096
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
def slot():
    ctx = xpng_amd.Context(W, W, 4, batch=B)
    blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    return dict(ctx=ctx, blobs=blobs, outs=outs, stream=torch.cuda.Stream())
a, b = slot(), slot()
rp = [r.data_ptr() for r in rs]
lens = a["ctx"].encode_device_batch(1, rp, [t.data_ptr() for t in a["blobs"]])
offs = [walk_tile_offsets(a["blobs"][i][:lens[i]].cpu().numpy().tobytes(), a["ctx"].n_tiles)[0] for i in range(B)]
b["ctx"].encode_device_batch(1, rp, [t.data_ptr() for t in b["blobs"]])
torch.cuda.synchronize()
def dec(s):
    s["ctx"].decode_device_batch(1, [t.data_ptr() for t in s["blobs"]], lens, offs, [t.data_ptr() for t in s["outs"]], stream=s["stream"].cuda_stream)
dec(a); dec(b); torch.cuda.synchronize()
for r in range(reps):
    dec(a)
    # enough of the other work to cover the decode (~70 ms)
    sb = b["stream"].cuda_stream
    if other == "transform":
        for _ in range(30): b["ctx"].transform_device_batch(rp, stream=sb)
    elif other == "encode":
        for _ in range(3): b["ctx"].encode_device_batch(1, rp, [t.data_ptr() for t in b["blobs"]], stream=sb, sync=False)
    elif other == "decode":
        dec(b)
    torch.cuda.synchronize()
print("done", other)
