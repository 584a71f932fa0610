#!/usr/bin/env python3
"""Wall-clock marks inside xpnghip_encode_tiles / xpnghip_decode_tiles for ONE 4096^2 RGBA image (probe build, XPNG_TRACE_API)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["XPNG_USE_PROBES_LIB"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch  # (before the first HIP call of the process: torch refuses to initialise its device layer afterwards)
torch.cuda.init()
from xpng_amd import api
from xpng_amd.synth import synth_raster
r = synth_raster("photo", 4096, 4096, len(sys.argv) < 2 or sys.argv[1] != "rgb")
b = api.encode_tiles(1, r)
api.decode_tiles(1, b, 4096, 4096, r.shape[2])
for i in range(3):
    t = time.perf_counter(); api.encode_tiles(1, r); e = time.perf_counter() - t
    t = time.perf_counter(); api.decode_tiles(1, b, 4096, 4096, r.shape[2]); d = time.perf_counter() - t
print("warm: encode %.2f ms decode %.2f ms" % (e * 1e3, d * 1e3))
os.environ["XPNG_TRACE_API"] = "1"
print("--- encode"); api.encode_tiles(1, r)
print("--- decode"); sys.stdout.flush(); api.decode_tiles(1, b, 4096, 4096, r.shape[2])

# ---- on-device decode of each pipeline shard ALONE and of the three together (own contexts, own streams)
import torch, xpng_amd
from xpng_amd.api import walk_tile_offsets
os.environ.pop("XPNG_TRACE_API")
W = H = 4096; ch = r.shape[2]
full = xpng_amd.Context(W, H, ch)
N = full.n_tiles
off, total = walk_tile_offsets(b, N); off.append(total)
d_blob = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
d_blob = torch.cat([d_blob, torch.zeros(64, dtype=torch.uint8, device="cuda")])
d_out = torch.zeros(W * H * ch + 64, dtype=torch.uint8, device="cuda")
rows = sorted(set(t[1] for t in full.tiles()))
row_start = [min(i for i, t in enumerate(full.tiles()) if t[1] == y) for y in rows] + [N]
R = len(rows); mid = 1 + (R - 1) // 2
ranges = [(row_start[0], row_start[1]), (row_start[1], row_start[mid]), (row_start[mid], row_start[R])]
ctxs = [xpng_amd.Context(W, H, ch, tile_range=rg) for rg in ranges]
streams = [torch.cuda.Stream() for _ in ranges]
def run(k):
    t0, t1 = ranges[k]
    ctxs[k].decode_device(1, d_blob.data_ptr() + off[t0], off[t1] - off[t0], [o - off[t0] for o in off[t0:t1]], d_out.data_ptr(), t0, t1, stream=streams[k].cuda_stream)
def timed(ks, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in ks: run(k)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
for k in range(3): run(k)
torch.cuda.synchronize()
for k in range(3): print("shard", k, ranges[k], "alone: %.2f ms" % timed([k]))
print("three together: %.2f ms" % timed([0, 1, 2]))
print("1 + 2 together: %.2f ms" % timed([1, 2]))
