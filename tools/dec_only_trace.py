#!/usr/bin/env python3
"""Batched encode once, then a few batched decodes (no verification): for kernel-trace experiments.  usage: dec_only_trace.py [B=64] [reps=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import xpng_amd
from xpng_amd.api import walk_tile_offsets
from xpng_amd.synth import synth_raster_torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
r = synth_raster_torch("photo", W, W, True, seed=1)
ctx = xpng_amd.Context(W, W, 4, batch=B)
blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
lens = ctx.encode_device_batch(1, [r.data_ptr()] * B, [b.data_ptr() for b in blobs])
off = walk_tile_offsets(blobs[0][:lens[0]].cpu().numpy().tobytes(), ctx.n_tiles)[0]
for _ in range(reps):
    ctx.decode_device_batch(1, [b.data_ptr() for b in blobs], lens, [off] * B, [o.data_ptr() for o in outs])
torch.cuda.synchronize()
print("ok", bool(torch.equal(outs[0][: W * W * 4], r.reshape(-1))))
