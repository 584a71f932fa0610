#!/usr/bin/env python3
"""Which kind of neighbour slows the serial-chain path?  One context encodes / decodes a 64-image batch strictly serially
(its time is the sum of its chain kernels' latencies) while ONE synthetic load (tools/ubench/hammer.hip) keeps the rest of the
GPU busy on another stream: VALU issue, LDS, streaming HBM copy, scattered 16-byte HBM rows, instruction-cache footprint.
Prints ms per encode and per decode, alone and beside each load.
usage: interference.py [B=64] [loads=none,valu,lds,stream,scatter,icache]"""
import ctypes, os, subprocess, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch

so = os.path.join(ROOT, "tools", "ubench", "hammer.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "ubench", "hammer.hip"), "-o", so])
H = ctypes.CDLL(so)
vp, u64, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
H.hammer_valu.argtypes = H.hammer_lds.argtypes = H.hammer_icache.argtypes = [vp, vp, ci, ci]
H.hammer_stream.argtypes = [vp, vp, vp, u64, ci]
H.hammer_scatter.argtypes = [vp, vp, vp, u64, ci, ci]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
loads = (sys.argv[2] if len(sys.argv) > 2 else "none,valu,lds,stream,scatter,icache").split(",")
W = 4096
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
ctx = xpng_amd.Context(W, W, 4, batch=B)
blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
rp, bp, op = [r.data_ptr() for r in rs], [t.data_ptr() for t in blobs], [t.data_ptr() for t in outs]
main = torch.cuda.Stream()
hs = torch.cuda.Stream()
lens = ctx.encode_device_batch(1, rp, bp, stream=main.cuda_stream)
ctx.decode_device_batch(1, bp, lens, None, op, stream=main.cuda_stream)
torch.cuda.synchronize()
assert ctx.decode_status() == 0 and all(torch.equal(outs[i][: W * W * 4], rs[i].reshape(-1)[: W * W * 4]) for i in range(0, B, 7))

scratch = torch.zeros(1024, dtype=torch.int32, device="cuda")
big_a = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
big_b = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
hsp = hs.cuda_stream


def launch(kind):
    if kind == "valu": H.hammer_valu(hsp, scratch.data_ptr(), 256 * 8, 4000)
    elif kind == "lds": H.hammer_lds(hsp, scratch.data_ptr(), 256 * 8, 600)
    elif kind == "stream": H.hammer_stream(hsp, big_a.data_ptr(), big_b.data_ptr(), 2 << 30, 256 * 16)
    elif kind == "scatter": H.hammer_scatter(hsp, big_a.data_ptr(), big_b.data_ptr(), 2 << 30, 256 * 10, 1)
    elif kind == "icache": H.hammer_icache(hsp, scratch.data_ptr(), 256 * 8, 40)


def hammer_ms(kind):
    launch(kind); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(hs):
        e0.record(); launch(kind); launch(kind); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2


def timed(fn, kind, per, cover_ms):
    n_h = 0 if kind == "none" else max(1, int(cover_ms / per) + 1)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(n_h): launch(kind)
        time.sleep(0.002)  # the load is running before the measured call is queued
        with torch.cuda.stream(main):
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


enc = lambda: ctx.encode_device_batch(1, rp, bp, stream=main.cuda_stream, sync=False)
dec = lambda: ctx.decode_device_batch(1, bp, lens, None, op, stream=main.cuda_stream)
for _ in range(3): enc(); dec()
torch.cuda.synchronize()
base = {}
for kind in loads:
    per = 1.0 if kind == "none" else hammer_ms(kind)
    e = timed(enc, kind, per, 250.0)
    d = timed(dec, kind, per, 300.0)
    if kind == "none": base = dict(e=e, d=d)
    print(f"{kind:8s} load launch {per:7.2f} ms | encode {e:7.2f} ms ({e / base.get('e', e):.2f}x) | decode {d:7.2f} ms ({d / base.get('d', d):.2f}x)", flush=True)
