#!/usr/bin/env python3
"""What does each kernel COST in the pipelined steady state?  Runs the bench's encode+decode step (P slots x B rasters, no
verification) once per kernel with that kernel left unlaunched (XPNG_SKIP, csrc/common.hpp: the workspaces keep the previous,
identical results, so everything downstream still works on valid data) and prints the step time beside the full pipeline's.
The difference is the kernel's marginal cost where it matters - isolated durations say little about a pipeline in which
latency-bound chains and bandwidth kernels of several slots overlap.
usage: knockout.py [P=5] [B=64] [mode=both|enc|dec] [names,comma,separated (a+b = both together) | all]          (child: knockout.py --child ...)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["chooser", "transform", "streams", "prep_a", "chain_a", "prep_c", "chain_c", "finish", "gather",
         "dec_prep", "dec_chain_a", "dec_alpha", "dec_chain_c", "dec_odd", "walk_small", "walk_big",
         "resid_small", "recon_small", "resid_big", "recon_big"]
GROUPS = {"enc_chains": "chain_a,chain_c", "dec_chains": "dec_chain_a,dec_chain_c,walk_small,walk_big",
          "all_chains": "chain_a,chain_c,dec_chain_a,dec_chain_c,walk_small,walk_big",
          "enc_bw": "chooser,transform,streams,prep_a,prep_c,finish,gather",
          "dec_bw": "dec_prep,dec_alpha,dec_odd,resid_small,recon_small,resid_big,recon_big",
          "all_bw": "chooser,transform,streams,prep_a,prep_c,finish,gather,dec_prep,dec_alpha,dec_odd,resid_small,recon_small,resid_big,recon_big"}

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    P, B, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    os.environ["XPNG_USE_PROBES_LIB"] = "1"  # the switches this tool uses exist only in libxpng_hip_probes.so (make probes)
    sys.path.insert(0, ROOT)
    import torch
    import xpng_amd
    from xpng_amd.synth import synth_raster_torch
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
    for sl in slots:  # two complete, serial passes per slot: every workspace holds valid results before anything is skipped
        for _ in range(2):
            lens = sl["ctx"].encode_device_batch(1, rp, sl["bp"])
            sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"])
    torch.cuda.synchronize()

    def step(k):
        sl = slots[k % P]
        sh = sl["stream"].cuda_stream
        if mode != "dec": sl["ctx"].encode_device_batch(1, rp, sl["bp"], stream=sh, sync=False)
        if mode != "enc": sl["ctx"].decode_device_batch(1, sl["bp"], lens, None, sl["op"], stream=sh)
    for k in range(2 * P): step(k)
    torch.cuda.synchronize()
    stagger_ms = float(os.environ.get("KNOCKOUT_STAGGER_MS", "0"))  # slot i starts i x this much later (phase experiment)
    if stagger_ms:
        for i, sl in enumerate(slots):
            with torch.cuda.stream(sl["stream"]):
                torch.cuda._sleep(int(i * stagger_ms * 2.4e6))
    steps = int(os.environ.get("KNOCKOUT_STEPS", "20"))
    # KNOCKOUT_NOISE=n: n one-element kernels on a stream of their own behind every step's launches (do kernel boundaries - their cache
    # write-backs and invalidations - slow the kernels in flight?)
    noise = int(os.environ.get("KNOCKOUT_NOISE", "0"))
    nstream, nbuf = torch.cuda.Stream(), torch.zeros(64, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
        if noise:
            with torch.cuda.stream(nstream):
                for _ in range(noise): nbuf.add_(1)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f}")
    sys.exit(0)

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
sel = sys.argv[4] if len(sys.argv) > 4 else "all"
def expand(case):  # "all_bw+chain_a": several names or groups knocked out together
    return ",".join(GROUPS.get(n, n) for n in case.split("+"))
cases = [("none", "")] + ([(n, n) for n in NAMES] + list(GROUPS.items()) if sel == "all" else [(n, expand(n)) for n in sel.split(",")])
base = None
for name, skip in cases:
    env = dict(os.environ, XPNG_SKIP=skip or "nothing", XPNG_SKIP_AFTER=str(4 * P))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(P), str(B), mode], env=env, capture_output=True, text=True, timeout=300)
    try:
        ms = float(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(name, "FAILED", r.stderr[-300:], flush=True)
        continue
    if base is None: base = ms
    print(f"{name:24s} {ms:8.2f} ms/step   saves {base - ms:6.2f} ms ({(base - ms) / base * 100:5.1f} %)", flush=True)
