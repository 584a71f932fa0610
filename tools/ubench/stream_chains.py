#!/usr/bin/env python3
"""Do kernels QUEUED BEHIND each other on N streams run N at a time?  Each of N streams gets K back-to-back 1-workgroup spin kernels
of ~3 ms (tools/ubench/hammer.hip k_valu); ideal wall time = K x 3 ms for every N.  Second form: stream pairs with a fork and a join
per kernel pair (event record / wait), the shape of a pipeline slot (caller's stream + side stream).
usage: stream_chains.py [GPU_MAX_HW_QUEUES]"""
import ctypes, os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1] if len(sys.argv) > 1 else "32"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
H = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "hammer.so"))
H.hammer_valu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
scratch = torch.zeros(64, dtype=torch.int32, device="cuda")
def spin(s, iters, blocks=1): H.hammer_valu(s.cuda_stream, scratch.data_ptr(), blocks, iters)
def warm():
    spin(torch.cuda.current_stream(), 20000, 2048); torch.cuda.synchronize()
warm()
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
s0 = torch.cuda.Stream()
one = min(timed(lambda: spin(s0, 20000)) for _ in range(3))
it = int(20000 * 3.0 / one)
one = min(timed(lambda: spin(s0, it)) for _ in range(3))
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"], f"one kernel {one:.2f} ms")
K = 4
for n in (1, 2, 4, 6, 8, 12, 16, 24):
    streams = [torch.cuda.Stream() for _ in range(n)]
    def plain():
        for k in range(K):
            for s in streams: spin(s, it)
    warm(); t = min(timed(plain) for _ in range(3))
    print(f"plain  N={n:3d} streams x {K} kernels: {t:7.2f} ms (ideal {K * one:5.1f})  -> kernels running at once ~{n * K * one / t:5.1f}")
for n in (2, 4, 8, 12):   # n slots = 2 n streams
    mains = [torch.cuda.Stream() for _ in range(n)]; sides = [torch.cuda.Stream() for _ in range(n)]
    evf = [torch.cuda.Event() for _ in range(n)]; evj = [torch.cuda.Event() for _ in range(n)]
    def forked():
        for k in range(K):
            for i in range(n):
                evf[i].record(mains[i]); sides[i].wait_event(evf[i])
                spin(mains[i], it); spin(sides[i], it)
                evj[i].record(sides[i]); mains[i].wait_event(evj[i])
    warm(); t = min(timed(forked) for _ in range(3))
    print(f"forked N={n:3d} slots (2 streams each) x {K} kernel pairs: {t:7.2f} ms (ideal {K * one:5.1f})  -> kernels running at once ~{2 * n * K * one / t:5.1f}")
# the same with 162-workgroup kernels (a chain launch's size)
for n in (4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(n)]
    def plain162():
        for k in range(K):
            for s in streams: spin(s, it, 162)
    warm(); t = min(timed(plain162) for _ in range(3))
    print(f"plain  N={n:3d} streams x {K} kernels of 162 workgroups: {t:7.2f} ms (ideal {K * one:5.1f})  -> at once ~{n * K * one / t:5.1f}")
