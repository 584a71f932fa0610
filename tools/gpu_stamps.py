#!/usr/bin/env python3
"""Phase timing of the serial-chain kernels from in-kernel s_memtime stamps (XPNG_STAMPS=1 debug build path)."""
import os, sys
import numpy as np
os.environ["XPNG_STAMPS"] = "1"
os.environ["XPNG_USE_PROBES_LIB"] = "1"  # the switches this tool uses exist only in libxpng_hip_probes.so (make probes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.api import walk_tile_offsets
from xpng_amd.synth import synth_raster_torch

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d_r = synth_raster_torch("photo", W, H, True)
ctx = xpng_amd.Context(W, H, 4)
d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
blobs = d_b[:n].cpu().numpy().tobytes()
off, _ = walk_tile_offsets(blobs, ctx.n_tiles)
d_o = torch.zeros(W * H * 4 + 64, dtype=torch.uint8, device="cuda")
ctx.decode_device(1, d_b.data_ptr(), n, off, d_o.data_ptr())
torch.cuda.synchronize()
for ti in (0, 1, 40):
    if ti >= ctx.n_tiles: continue
    e = ctx.fetch(40, ti, 640).view(np.uint64).reshape(10, 8).astype(np.int64)
    d = ctx.fetch(41, ti, 640).view(np.uint64).reshape(10, 8).astype(np.int64)
    t = ctx.tile(ti)
    print(f"tile {ti} {t} n={t[2]*t[3]}")
    for c in (0, 2, 3, 9):
        en = e[c]
        print(f"  enc stream {c}: hist {en[1]-en[0]:>9d}  tables {en[2]-en[1]:>9d}  recurrence {en[3]-en[2]:>10d}  hdr/table {en[4]-en[3]:>8d}  rawchk {en[5]-en[4]:>8d}  (cycles)")
        dn = d[c]
        print(f"  dec stream {c}: setup {dn[1]-dn[0]:>9d}  recurrence {dn[2]-dn[1]:>10d}")
