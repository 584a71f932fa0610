#!/usr/bin/env python3
"""How many kernels of different HIP streams really run at once?  N streams each get one 1-workgroup spin kernel of ~5 ms
(tools/ubench/hammer.hip k_valu, a fixed instruction count); wall time of the batch / 5 ms = serialisation factor.
usage: stream_concurrency.py [GPU_MAX_HW_QUEUES]"""
import ctypes, os, sys, time
if len(sys.argv) > 1: os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
H = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "hammer.so"))
H.hammer_valu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
scratch = torch.zeros(64, dtype=torch.int32, device="cuda")
def run(n, iters, blocks=1):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams: H.hammer_valu(s.cuda_stream, scratch.data_ptr(), blocks, iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
def warm():  # clocks up: ~60 ms of full-GPU work
    H.hammer_valu(torch.cuda.current_stream().cuda_stream, scratch.data_ptr(), 2048, 20000)
    torch.cuda.synchronize()
warm()
it = 20000
one = min(run(1, it) for _ in range(3))
it = int(it * 5.0 / one)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for n in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    warm()
    one = min(run(1, it) for _ in range(2))
    t = min(run(n, it) for _ in range(3))
    print(f"N={n:3d} streams: one {one:6.2f} ms, batch {t:7.2f} ms  -> concurrency ~{n * one / t:5.1f}")
