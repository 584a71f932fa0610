#!/usr/bin/env python3
"""Does capturing one slot's step (batched encode + decode: ~35 kernels on four streams joined by events) in a hipGraph shorten it?
usage: graph_try.py [P=4] [B=64] [steps=20]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xpng_amd
from xpng_amd.synth import synth_raster_torch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W = 4096
rs = [synth_raster_torch("photo", W, W, True, seed=1 + b) for b in range(B)]
rp = [r.data_ptr() for r in rs]
slots = []
for p in range(P):
    ctx = xpng_amd.Context(W, W, 4, batch=B)
    blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    outs = [torch.zeros(W * W * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    slots.append(dict(ctx=ctx, bp=[t.data_ptr() for t in blobs], op=[t.data_ptr() for t in outs], stream=torch.cuda.Stream(), keep=(blobs, outs)))
lens = None
for sl in slots:
    lens = sl["ctx"].encode_device_batch(1, rp, sl["bp"])
    sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"])
torch.cuda.synchronize()
def step_stream(sl):
    sh = sl["stream"].cuda_stream
    sl["ctx"].encode_device_batch(1, rp, sl["bp"], stream=sh, sync=False)
    sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"], stream=sh)
def run(fn, n):
    for k in range(2 * P): fn(slots[k % P])
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(n): fn(slots[k % P])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("streams: %.2f ms per step" % run(step_stream, steps))
# capture every slot's step once
for sl in slots:
    step_stream(sl)
torch.cuda.synchronize()
ok = True
for sl in slots:
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=sl["stream"], capture_error_mode="relaxed"):
            step_stream(sl)
        sl["graph"] = g
    except Exception as ex:
        print("capture failed:", repr(ex)[:300]); ok = False; break
if ok:
    def step_graph(sl):
        with torch.cuda.stream(sl["stream"]):
            sl["graph"].replay()
    print("graphs : %.2f ms per step" % run(step_graph, steps))
    # the outputs are still right
    torch.cuda.synchronize()
    print("decode equals source:", all(bool(torch.equal(slots[0]["keep"][1][b][: W * W * 4].view(W, W, 4), rs[b])) for b in range(0, B, 17)))
