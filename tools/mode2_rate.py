#!/usr/bin/env python3
"""Level-2 (rANS v1, RGB) batched encode+decode rate, B distinct rasters, serial steps.  usage: mode2_rate.py [B=16] [steps=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.api import walk_tile_offsets
from xpng_amd.synth import synth_raster_torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = 4096
rs = [synth_raster_torch("photo", W, W, False, seed=1 + b) for b in range(B)]
ctx = xpng_amd.Context(W, W, 3, batch=B)
blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
outs = [torch.zeros(W * W * 3 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
for mode in (1, 2):
    lens = ctx.encode_device_batch(mode, [r.data_ptr() for r in rs], [b.data_ptr() for b in blobs])
    offs = [walk_tile_offsets(blobs[i][:lens[i]].cpu().numpy().tobytes(), ctx.n_tiles)[0] for i in range(B)]
    ctx.decode_device_batch(mode, [b.data_ptr() for b in blobs], lens, offs, [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    ok = all(bool(torch.equal(outs[i][: W * W * 3], rs[i].reshape(-1))) for i in range(B))
    def t(fn):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
    te = t(lambda: ctx.encode_device_batch(mode, [r.data_ptr() for r in rs], [b.data_ptr() for b in blobs], sync=False))
    td = t(lambda: ctx.decode_device_batch(mode, [b.data_ptr() for b in blobs], lens, offs, [o.data_ptr() for o in outs]))
    px = B * W * W
    print(f"level {mode} RGB B={B}: roundtrip {ok}  bytes/img {lens[0]}  encode {te*1e3:.1f} ms ({px/te/1e9:.2f} Gpx/s)  decode {td*1e3:.1f} ms ({px/td/1e9:.2f} Gpx/s)")
