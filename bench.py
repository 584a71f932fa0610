#!/usr/bin/env python3
"""bench.py -- xPNG hot path on MI355X: Mpixels/s encode+decode, bit-exact vs the reference.

  python bench.py [--gpus N --steps K --warmup W]

N > 1 is one process per GPU over torch.distributed (RCCL).  Run under `python -m torch.distributed.run --nproc-per-node N ...`
the script is rank RANK of WORLD_SIZE; run PLAIN with --gpus N > 1 (no WORLD_SIZE in the environment) it starts the N ranks
itself - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a child process, before
anything here touches the GPU - relays rank 0's JSON line and exits with the child's code.  A WORLD_SIZE that is not N is an
error: the line never reports an n_gpus other than the one asked for.

A step = one pass of the hot path over one batch of synthetic rasters already resident in HBM:
    tile ENCODE (rasters -> concatenated tile blobs)  [+ RCCL exchange of blob ranges when N>1]
  + tile DECODE (blobs -> rasters; the tile-size walk of libxpng.c:982 included, on the device).
Headline workload (`value`): each rank owns a 4096x4096-pixel share of a synthetic `photo` RGBA raster,
level -1; at N=1 that is BASELINE.json's "4096x4096 synthetic RGBA8, level -1" configuration, at N ranks the global raster is
(4096*a)x(4096*b), a*b=N, cut into contiguous tile ranges (weak scaling; tiles are independent, reference libxpng.c:542-570).

The same JSON line also carries, each measured by the same code path (`run_leg`):
  legs.rgb_l1 / legs.rgb_l2   the other two legs of BASELINE config 3 (4096^2 `photo` RGB at -1 and at -2), N=1 only
  config4                     BASELINE config 4: ONE fixed 16384^2 RGBA raster cut over the N ranks (strong scaling)
  single_image                latency of ONE 4096^2 image: on-device, and through the host-buffer C entry points that
                              xpng_store / xpng_load call (PCIe included), next to the reference's CPU run of the same call
Before timing, every leg is verified: md5(header + blobs) against the reference-generated manifest where pinned, and
decode(encode(x)) == x for every image.  `roofline` is the bandwidth-bound kernel pair of the path (predictor chooser +
per-pixel transform, BASELINE config 2), timed live with HIP events on its stream; `cpu_baseline` is the compiled reference
(oracle/_ref/xpng, kind "reference") or the oracle's C port, timed on this node's host cores.
"""
import argparse
import hashlib
import json
import os

# Every pipeline slot drives four HIP streams (main, alpha-encode side branch, alpha-decode side branch, small-tile decode
# tail).  The HIP runtime maps streams onto 4 hardware queues by default, and streams that share a queue serialise; this
# must be set before the runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_PX = {4: 10.0, 3: 7.75}  # read PXSZ + chooser re-read PXSZ/4 + write PXSZ+1
PMC_SUFFIX = {4: "pmc_transform_rgba.json", 3: "pmc_transform_rgb.json"}  # newest profiles/rNN_<suffix>


# Environment hygiene.  The release library reads only same-bytes form selectors; every XPNG_* variable
# seen is printed in `config.env`.  The timing-study switches (kernel knock-outs, no-store, pads ...) exist only in
# libxpng_hip_probes.so, which this script loads only under --probe-run (tools/): such a line carries "probe_run": true and is
# not a benchmark.  Without that flag a skip / no-store switch or the probe library in the environment is refused outright.
FALSIFYING = ("XPNG_SKIP", "XPNG_SKIP_AFTER", "XPNG_DBG_NOSTORE", "XPNG_USE_PROBES_LIB", "XPNG_FAKE_DEVICES")


def env_report(probe_run):
    seen = {k: v for k, v in sorted(os.environ.items()) if k.startswith("XPNG_")}
    bad = [k for k in seen if k in FALSIFYING or k.startswith("XPNG_PAD_") or k.startswith("XPNG_DBG_")]
    if bad and not probe_run:
        raise SystemExit(f"bench.py: refusing to run with timing-study switches in the environment: {bad} (tools pass --probe-run; such a run is not a benchmark)")
    if probe_run:
        os.environ["XPNG_USE_PROBES_LIB"] = "1"
        seen["XPNG_USE_PROBES_LIB"] = "1"
    return seen


def grid_for(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2), 16: (4, 4)}.get(n, (n, 1))


def cpu_baseline(raster_np, level=1, budget_s=20.0):
    """Reference (or port) encode+decode of the same raster on the host cores.  rank 0, N=1 only."""
    from xpng_amd.synth import to_seven_bytes
    h, w, ch = raster_np.shape
    px = w * h
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "xpng")
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    if os.access(ref_bin, os.X_OK):
        best_e = best_d = 0.0
        threads = None
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            src, dst, back = os.path.join(td, "in.7"), os.path.join(td, "out.xpng"), os.path.join(td, "out.7")
            with open(src, "wb") as f:
                f.write(to_seven_bytes(raster_np))
            t_end = time.time() + budget_s
            runs = 0
            while runs < 7 and (runs < 3 or time.time() < t_end):
                e = subprocess.run([ref_bin, f"-{level}", src, dst], capture_output=True, text=True)
                d = subprocess.run([ref_bin, "-d", dst, back], capture_output=True, text=True)
                me = re.search(r"encode,\s+(\d+) thread.?:\s+(\d+) MPx/s", e.stdout)
                md = re.search(r"decode,\s+(\d+) thread.?:\s+(\d+) MPx/s", d.stdout)
                if not (me and md):
                    break
                threads = int(me.group(1))
                best_e, best_d = max(best_e, float(me.group(2))), max(best_d, float(md.group(2)))
                runs += 1
        if best_e and best_d:
            return {"value": round(1.0 / (1.0 / best_e + 1.0 / best_d), 1), "unit": "Mpx/s", "cores": threads, "kind": "reference",
                    "encode_mpx_s": best_e, "decode_mpx_s": best_d, "host_cpus": cores,
                    "encode_ms": round(px / best_e / 1e3, 2), "decode_ms": round(px / best_d / 1e3, 2),
                    "sample": f"compiled reference libxpng.c (build.sh flags), xpng -{level} / -d on ONE {w}x{h}x{ch} raster of the workload, "
                              f"best of {runs} runs, T=min(tiles,nproc); timed region = the reference's own (libxpng.c:727-760, 967-985: memory to memory, normalize_RGBA included, file I/O excluded)"}
    from oracle import pyoracle as po  # CPU port as the fallback baseline
    best_e = best_d = 0.0
    for _ in range(3):
        data = po.encode_image(level, raster_np)
        best_e = max(best_e, px / (po.last_encode_ns() / 1e9) / 1e6)
        po.decode_image(data)
        best_d = max(best_d, px / (po.last_decode_ns() / 1e9) / 1e6)
    return {"value": round(1.0 / (1.0 / best_e + 1.0 / best_d), 1), "unit": "Mpx/s", "cores": min(cores, 81), "kind": "port",
            "encode_mpx_s": round(best_e, 1), "decode_mpx_s": round(best_d, 1), "host_cpus": cores,
            "encode_ms": round(px / best_e / 1e3, 2), "decode_ms": round(px / best_d / 1e3, 2),
            "sample": f"oracle C port, level {level} encode+decode of ONE {w}x{h}x{ch} raster of the workload, best of 3"}


def ctx_len_at(ctx, i):
    from xpng_amd.api import hip_lib
    return hip_lib().xpnghip_ctx_last_blobs_len_at(ctx._h, i)


def spawn_ranks(args, argv):
    """--gpus N > 1 run plainly (no WORLD_SIZE): start the N ranks as a CHILD torch.distributed.run (never an exec: this process may
    not have touched the GPU yet, but a launcher must not rely on that), relay its output, return its exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    ok = False
    for ln in out.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, flush=True)
    if lines:
        try:
            ok = json.loads(lines[-1]).get("n_gpus") == args.gpus
        except Exception:
            ok = False
        print(lines[-1], flush=True)
    if out.returncode == 0 and not ok:
        print(f"bench.py: the {args.gpus}-rank child did not report n_gpus == {args.gpus}", file=sys.stderr)
        return 3
    return out.returncode


def newest_profile(pattern):
    """newest committed profiles/rNN_*<pattern> (by round number, then name)"""
    import glob
    best = None
    for fn in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*" + pattern)):
        key = (os.path.basename(fn)[:3], os.path.basename(fn))
        if best is None or key > best[0]:
            best = (key, fn)
    return best[1] if best else None


def step_roofline_obj(res):
    """The WHOLE step against the HBM roofline: algorithmic bytes of an encode + decode of one pixel = read PXSZ +
    write the compressed bytes (encode), read them + write PXSZ (decode): SURVEY.md 8(d); achieved = that x pixels per step /
    ms_per_step.  traffic_* = what the counters saw for one step (newest profiles/*pmc_step_mem*.json of this pixel format)."""
    ch, level = res["ch"], res["level"]
    px_img = res["W"] * res["H"]
    algo = 2.0 * ch + 2.0 * res["compressed_bytes"] / px_img
    achieved = algo * res["B"] * res["my_px"] / (res["ms_per_step"] * 1e-3) / 1e9
    out = {"bound": "hbm", "algorithmic_bytes_per_px": round(algo, 3), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic_bytes_per_px": None, "traffic_over_algorithmic": None, "traffic_source": None}
    fn = newest_profile("pmc_step_mem.json" if ch == 4 and level == 1 else ("pmc_step_mem_rgb.json" if level == 1 else "pmc_step_mem_rgb_l2.json"))
    if fn:
        try:
            pmc = json.load(open(fn))
            out["traffic_bytes_per_px"] = pmc["bytes_per_px"]
            out["traffic_over_algorithmic"] = round(pmc["bytes_per_px"] / algo, 2)
            out["traffic_source"] = "profiles/" + os.path.basename(fn) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over one step, serialised passes)"
        except Exception:
            pass
    return out


class Env:
    """process-wide state of one bench run (rank, device, backend)"""
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: refusing to print a line for another rank count")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the xPNG tile codec has no CPU fallback")
        self.local_rank = local_rank % torch.cuda.device_count()  # (a rehearsal may run several ranks on one GPU)
        torch.cuda.set_device(self.local_rank)
        self.dev = f"cuda:{self.local_rank}"
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(args.backend)
        self.use_host = self.world > 1 and args.backend != "nccl"

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def run_leg(env, W, H, alpha, level, B, P, steps, warmup, kind="photo", roofline_reps=0, single=False, stagger_ms=0.0):
    """One workload through the pipelined hot path: verify, (roofline pair), W warmup + K timed steps, re-verify.
    Returns a dict of measurements (rank 0 fills the reference checks)."""
    import xpng_amd
    from xpng_amd.shard import band_rows, exchange_blobs_round_robin, image_from_round_robin, tile_table, weighted_tile_ranges
    from xpng_amd.synth import seven_header, synth_raster_torch
    torch, dist, world, rank = env.torch, env.dist, env.world, env.rank
    ch = 4 if alpha else 3
    tiles = tile_table(W, H)
    t0, t1 = weighted_tile_ranges(tiles, world)[rank]
    # one context per pipeline slot, B images per launch, workspace only for this rank's tile range
    ctx = xpng_amd.Context(W, H, ch, device=env.local_rank, batch=B, tile_range=(t0, t1))
    assert ctx.tiles() == tiles
    y0, y1 = band_rows(tiles, t0, t1)
    # The rank materialises only the raster band its tiles touch.  The B rasters of a launch are DISTINCT (seeds 1..B; image 0
    # is the one the manifest pins): identical rasters would sit in the 256 MB Infinity Cache for the roofline kernels and
    # would make every lane of the wide entropy kernels take the same branches.
    # (+ one spare row when the band is not the tail of the image: the staged 16-byte loads of the band's last row may run a few
    # bytes past it - the kernels clamp at the end of the whole raster, which a band in the middle does not reach)
    band_stores = [synth_raster_torch(kind, W, min(H, y1 + 1) - y0, alpha, seed=1 + b, y0=y0, device=env.dev) for b in range(B)]
    bands = [bs[: y1 - y0] for bs in band_stores]
    band = bands[0]
    bpr = W * ch
    stream = torch.cuda.current_stream().cuda_stream
    rast_ptrs = [bd.data_ptr() - y0 * bpr for bd in bands]  # kernels address rows absolutely; only [y0, y1) is ever touched
    slots = []
    for i in range(max(1, P)):
        bl = [torch.empty(ctx.blob_bound(t0, t1) + 64, dtype=torch.uint8, device=env.dev) for _ in range(B)]
        bk = [torch.zeros_like(band) for _ in range(B)]
        slots.append(dict(ctx=ctx if i == 0 else xpng_amd.Context(W, H, ch, device=env.local_rank, batch=B, tile_range=(t0, t1)),
                          stream=torch.cuda.Stream(), blobs=bl, back=bk, blob_ptrs=[t.data_ptr() for t in bl],
                          back_ptrs=[t.data_ptr() - y0 * bpr for t in bk]))
    d_blobs_all, d_back_all, blob_ptrs, back_ptrs = slots[0]["blobs"], slots[0]["back"], slots[0]["blob_ptrs"], slots[0]["back_ptrs"]
    my_px = sum(t[2] * t[3] for t in tiles[t0:t1])
    total_px = W * H

    def same_tiles(a, b):  # rows shared with a neighbouring rank's tiles are not written by this rank: compare tile by tile
        if world == 1:
            return bool(torch.equal(a, b))
        return all(bool(torch.equal(a[ty - y0:ty - y0 + th, tx:tx + tw], b[ty - y0:ty - y0 + th, tx:tx + tw])) for (tx, ty, tw, th) in tiles[t0:t1])

    # ---- correctness gate (untimed): bit-exact vs reference manifest where pinned (image 0), round trip for every image
    n = ctx.encode_device(level, rast_ptrs[0], blob_ptrs[0], t0, t1, stream=stream)
    blob0 = d_blobs_all[0][:n].clone()
    lens_b = ctx.encode_device_batch(level, rast_ptrs, blob_ptrs, t0, t1, stream=stream)   # all B images, one launch sequence
    ok = lens_b[0] == n and bool(torch.equal(d_blobs_all[0][:n], blob0))                    # batched == single-image launch
    ctx.decode_device_batch(level, blob_ptrs, lens_b, None, back_ptrs, t0, t1, stream=stream)   # (tile sizes walked on the device)
    torch.cuda.synchronize()
    ok = ok and ctx.decode_status() == 0
    for bi in range(B):
        ok = ok and same_tiles(d_back_all[bi], bands[bi])
    verified = {"roundtrip": bool(ok), "images_verified": B}
    len_table = [None]

    def exchange(bufs, scratch=None):
        # the one exchange of the path: the file of image b is assembled on rank b % world, so every rank sends each other
        # rank ONE message per step (its tile-range blobs of that rank's images, packed): 56 messages over 56 directed xGMI
        # links on a node instead of 7 converging on rank 0.  RCCL send/recv, no collective on the data path.  The first call
        # learns the (world x B) length table; later calls pass it in, so nothing here synchronises with the host and the
        # exchange queues behind the encode on its stream
        if env.use_host:
            bufs = [t[:lens_b[i]].cpu() for i, t in enumerate(bufs)]
        recv, table = exchange_blobs_round_robin(bufs, lens_b, table=len_table[0], scratch=scratch)
        len_table[0] = table
        return recv, table

    if world > 1:
        recv, table = exchange(d_blobs_all)
        gathered, lens = (image_from_round_robin(recv, table, 0, 0) if rank == 0 else None), [row[0] for row in table]
    else:
        gathered, lens = d_blobs_all[0][:n], [n]
    if rank == 0:
        man_path = os.path.join(ROOT, "tests", "golden", "manifest.json")
        key = f"synth_{kind}_{W}x{H}_{'rgba' if alpha else 'rgb'}"
        if os.path.exists(man_path):
            man = json.load(open(man_path))
            if key in man and f"L{level}" in man[key]:
                md = hashlib.md5(seven_header(W, H, alpha, level=level) + gathered.cpu().numpy().tobytes()).hexdigest()
                verified["reference_size"] = 8 + int(sum(lens)) == man[key][f"L{level}"]["size"]
                verified["reference_md5"] = md == man[key][f"L{level}"]["md5"]
                ok = ok and verified["reference_md5"]
    if not ok:
        raise SystemExit(f"rank {rank}: {W}x{H} level {level}: output is NOT bit-exact / does not round-trip: {verified}")
    ref_blobs = [d_blobs_all[bi][:lens_b[bi]].clone() for bi in range(B)]  # what every slot must reproduce
    step_no = [0]

    def step():
        # one launch sequence covers all B images (virtual tile = image * N + tile); consecutive steps alternate pipeline slots
        sl = slots[step_no[0] % len(slots)]
        step_no[0] += 1
        sh = sl["stream"].cuda_stream
        sl["ctx"].encode_device_batch(level, rast_ptrs, sl["blob_ptrs"], t0, t1, stream=sh, sync=False)
        if world > 1:
            with torch.cuda.stream(sl["stream"]):
                exchange(sl["blobs"], sl.setdefault("xchg", {}))  # tile bytes are deterministic: the lengths are the ones verified above
        # decode as the reference times it (libxpng.c:967-985): the tile-size walk is part of the step, on the device
        sl["ctx"].decode_device_batch(level, sl["blob_ptrs"], lens_b, None, sl["back_ptrs"], t0, t1, stream=sh)

    def timed(fn, reps, warm=1, warm_s=0.0):  # HIP events on the stream the kernels run on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_w = time.perf_counter()
        k_w = 0
        while k_w < warm or time.perf_counter() - t_w < warm_s:  # (a leg may start on a GPU that idled through a CPU baseline: clocks down)
            fn(); torch.cuda.synchronize(); k_w += 1
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps  # ms

    res = {"W": W, "H": H, "ch": ch, "level": level, "B": B, "P": len(slots), "tiles": len(tiles), "tiles_per_rank": t1 - t0, "my_px": my_px,
           "compressed_bytes": int(sum(lens)), "verified": verified}
    # the roofline kernels run over the whole batch per launch (a single 4096^2 pass is ~30 us, i.e. launch-bound); timed
    # here, before the pipelined steps, with nothing else on the device
    if roofline_reps:
        # (a few untimed launches first: a leg may start on a GPU that idled through the previous leg's CPU baseline)
        res["tr_ms"] = timed(lambda: ctx.transform_device_batch(rast_ptrs, t0, t1, stream=stream), roofline_reps, warm=4, warm_s=0.4)
    if single:
        # ONE image of the workload, strictly serial, nothing else on the device (measured before the pipelined steps heat the
        # part up: a chain-latency-bound launch sequence scales with the clock): on-device latency ...
        reps = max(5, steps // 2)
        res["enc_ms"] = timed(lambda: ctx.encode_device(level, rast_ptrs[0], blob_ptrs[0], t0, t1, stream=stream, sync=False), reps, warm=2, warm_s=0.3)
        res["dec_ms"] = timed(lambda: ctx.decode_device(level, blob_ptrs[0], n, None, back_ptrs[0], t0, t1, stream=stream), reps, warm=2, warm_s=0.3)
        if world == 1:
            # ... and the wall time of the host-buffer entry points that xpng_store / xpng_load call (include/xpng_hip.h:
            # xpnghip_encode_tiles / xpnghip_decode_tiles): host raster in, malloc()ed blobs out, PCIe both ways included
            from xpng_amd import api
            host_r = band.cpu().numpy()
            blobs_h = api.encode_tiles(level, host_r)
            assert blobs_h == d_blobs_all[0][:n].cpu().numpy().tobytes()
            best_e = best_d = 1e9
            import numpy as np
            for _ in range(5):
                # the C calls alone, as the host driver makes them: encode returns its malloc()ed buffer (copying it into a Python
                # object and freeing it are this script's business, outside the clock); decode fills a buffer the caller has just
                # malloc()ed - untouched pages, like xpng_load's (libxpng.c:974) - whose release is outside the clock too
                t_a = time.perf_counter(); mb = api.encode_tiles(level, host_r, copy=False); best_e = min(best_e, time.perf_counter() - t_a)
                assert mb.n == len(blobs_h)
                mb.free()
                back = np.empty((H, W, ch), dtype=np.uint8)
                t_a = time.perf_counter(); api.decode_tiles(level, blobs_h, W, H, ch, out=back); best_d = min(best_d, time.perf_counter() - t_a)
                ok_back = bool((back == host_r).all())
                del back
                assert ok_back
            res["api_enc_ms"], res["api_dec_ms"] = best_e * 1e3, best_d * 1e3
            # ... and what xpng_store itself runs on the caller's raw raster (host/xpng_api.c store_on_device): xpnghip_image_begin
            # (upload + normalize_RGBA on the device) -> xpnghip_image_encode_T -> xpnghip_image_end, memory to memory: the
            # reference's own timed region (libxpng.c:727-760: validation, normalize_RGBA, the tile stage; no file output)
            best_s = 1e9
            for _ in range(5):
                t_a = time.perf_counter(); mb = api.image_store(level, host_r); best_s = min(best_s, time.perf_counter() - t_a)
                assert mb.n == len(blobs_h) and mb.bytes() == blobs_h
                mb.free()
            res["store_ms"] = best_s * 1e3
            # the host-buffer calls keep their contexts (three pipeline shards, each with streams of its own) in the library's pool of
            # idle ones; give them back before the pipelined steps: HIP maps streams onto the 32 hardware queues as they are created,
            # and with more streams alive than queues the slots' streams start sharing queues, i.e. serialising (measured: 38.5 -> 35)
            api.hip_lib().xpnghip_shutdown()
    # every pipeline slot allocates its decode workspace on first use: touch each once before the W warmup steps, so that a
    # small W still leaves no allocation inside the timed region (these P untimed steps are in addition to the W requested)
    for _ in range(len(slots)):
        step()
    env.barrier()
    step_no[0] = 0
    for _ in range(warmup):
        step()
    env.barrier()
    t_start = time.perf_counter()
    for si in range(steps):
        step()
        if stagger_ms > 0 and si < len(slots) - 1:
            time.sleep(stagger_ms * 1e-3)  # (experiment, inside the timed region: the slots start their cycles this far apart instead of together)
    env.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # every image of every pipeline slot must have produced the verified bytes and raster
    for si, sl in enumerate(slots):
        for bi in range(B):
            if ctx_len_at(sl["ctx"], bi) != lens_b[bi] or not torch.equal(sl["blobs"][bi][:lens_b[bi]], ref_blobs[bi]) or not same_tiles(sl["back"][bi], bands[bi]):
                raise SystemExit(f"rank {rank}: slot {si} image {bi} differs from the verified image")
    res.update(elapsed=elapsed, ms_per_step=elapsed / steps * 1e3, mpx_s=B * total_px * steps / elapsed / 1e6,
               hbm_in_use_gb=round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30, 1))
    res["host_raster"] = band.cpu().numpy() if (world == 1 and rank == 0) else None
    for sl in slots:
        sl["ctx"].close()
    del slots, bands, band_stores, d_blobs_all, d_back_all, ref_blobs
    torch.cuda.empty_cache()
    return res


def roofline_obj(res):
    ch, B, my_px, tr_ms = res["ch"], res["B"], res["my_px"], res["tr_ms"]
    achieved = ALGO_BYTES_PER_PX[ch] * my_px * B / (tr_ms * 1e-3) / 1e9
    # HBM bytes per launch of the roofline kernels from the committed PMC passes (FETCH_SIZE doubled + WRITE_SIZE, per pixel):
    # counters cannot be collected inside this run, so the figure is the profile's bytes/px times this launch's pixels
    traffic, traffic_src = None, None
    for pmc_path in (newest_profile(PMC_SUFFIX[ch]),):
        fn = os.path.basename(pmc_path) if pmc_path else ""
        if fn and os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                traffic = int(pmc["per_pixel_bytes"]["total"] * B * my_px)
                traffic_src = f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/px of that run x pixels of this launch)"
                break
            except Exception:
                traffic, traffic_src = None, None
    return {"kernel": "k_chooser + k_m1_transform (predictor chooser + per-pixel transform, BASELINE config 2)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_px": ALGO_BYTES_PER_PX[ch], "ms_per_launch": round(tr_ms, 4),
            "transform_mpx_s": round(B * my_px / tr_ms / 1e3, 1), "images_per_launch": B,
            "read_only_frac_of_peak": round(ch * B * my_px / (tr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--image", type=int, default=0, help="headline workload = ONE fixed image x image raster cut over the ranks (strong scaling) instead of 4096^2 shares")
    ap.add_argument("--share", type=int, default=4096, help="per-rank share edge in pixels (weak scaling)")
    ap.add_argument("--rgb", action="store_true", help="headline workload RGB instead of RGBA")
    ap.add_argument("--level", type=int, default=1, choices=(1, 2), help="headline workload level (2 = SLOW, RGB only)")
    ap.add_argument("--kind", default="photo")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the RGB legs of config 3")
    ap.add_argument("--no-config4", action="store_true", help="skip the 16384^2 strong-scaling leg")
    ap.add_argument("--roofline-reps", type=int, default=50)
    ap.add_argument("--pipeline", type=int, default=8,
                    help="contexts (each with its own HIP stream and buffers) that consecutive steps alternate between, so the "
                         "encode of step k+1 runs beside the decode of step k; 1 = strictly serial steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo moves blobs through the host, for rehearsals)")
    ap.add_argument("--batch", type=int, default=64,
                    help="images per launch per rank: a step encodes+decodes `batch` rasters of the workload in one batched launch "
                         "sequence (the entropy stage is a serial chain per tile stream, so one image alone cannot fill 256 CUs)")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="experiment: host sleep between the first steps of the pipeline slots (warm-up), so that their cycles start apart; 0 = off")
    ap.add_argument("--probe-run", action="store_true", help="tools only: load libxpng_hip_probes.so and accept timing-study switches; the line is marked probe_run and is not a benchmark")
    ap.add_argument("--spawn-check", action="store_true", help="CPU rehearsal of the rank start-up only: the ranks rendezvous (gloo), agree on the world size and rank 0 prints a line with n_gpus and spawn_check: true; no GPU is touched and nothing is measured")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))  # (before anything initialises the GPU in this process)
    if args.spawn_check:
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        if world > 1:
            import torch
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            assert int(t.item()) == world
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "SPAWN CHECK, not a benchmark", "spawn_check": True, "n_gpus": world, "value": None}), flush=True)
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return
    env_seen = env_report(args.probe_run)
    if args.level == 2 and not args.rgb:
        raise SystemExit("--level 2 codes RGB only (libxpng.c:755 sends RGBA to level 1): add --rgb")

    env = Env(args)
    world, rank = env.world, env.rank
    alpha = not args.rgb
    if args.image:
        W = H = args.image
        scaling = "strong"
    else:
        a, b = grid_for(world)
        W, H = args.share * a, args.share * b
        scaling = "weak"
    B, P = max(1, args.batch), max(1, args.pipeline)

    main_res = run_leg(env, W, H, alpha, args.level, B, P, args.steps, args.warmup, kind=args.kind,
                       roofline_reps=args.roofline_reps, single=True, stagger_ms=args.stagger_ms)
    host_raster = main_res.pop("host_raster")
    cpu = cpu_baseline(host_raster, args.level) if (world == 1 and rank == 0 and not args.no_cpu) else None

    legs = {}
    if world == 1 and rank == 0 and not args.no_legs and not args.image and alpha and args.level == 1 and args.share == 4096:
        # The other two legs of BASELINE config 3: `photo` RGB at -1 and at -2.  Each runs as a CHILD process of
        # this script with its own flags (the same run_leg code path, its own reference CPU baseline): a leg measured in this
        # process after the 150 GB of the headline leg have been allocated and freed gets its buffers from fragmented device
        # memory and a down-clocked part, and its bandwidth-bound roofline pair reads 0.31-0.36 instead of the 0.55 of a clean run
        env.torch.cuda.empty_cache()
        for name, lvl in (("rgb_l1", 1), ("rgb_l2", 2)):
            # (level 2 holds 27 B/px of workspace against level 1's 12: 9/16 of the rasters per launch - 36 of 64 - keep it under 160 GB;
            #  profiles/r04_experiments.txt: 48 x 8 reads 18.6-19.0 Gpx/s at 210 GB, 36 x 8 17.4 at 158 GB, 24 x 8 17.2 at 106 GB)
            cmd = [sys.executable, os.path.abspath(__file__), "--rgb", "--level", str(lvl), "--batch", str(min(B, 128) if lvl == 1 else max(1, min(B, 128) * 9 // 16)), "--pipeline", str(min(P, 8)),
                   "--steps", str(max(6, args.steps // 3)), "--warmup", "2", "--roofline-reps", str(max(10, args.roofline_reps // 2)), "--no-legs", "--no-config4"]
            if args.no_cpu:
                cmd.append("--no-cpu")
            if args.probe_run:
                cmd.append("--probe-run")
            try:
                out_c = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
                line = [ln for ln in out_c.stdout.splitlines() if ln.startswith("{")][-1]
                c = json.loads(line)
                legs[name] = {"workload": c["config"]["workload"], "value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"],
                              "batch": c["config"]["batch"], "pipeline_slots": c["config"]["pipeline_slots"], "compressed_bytes": c["config"]["compressed_bytes"],
                              "verified": c["verified"], "hbm_in_use_gb": c["config"]["hbm_in_use_gb"],
                              "single_image_encode_ms": c["single_image_encode_ms"], "single_image_decode_ms": c["single_image_decode_ms"],
                              "single_image": c["single_image"], "roofline": c["roofline"], "step_roofline": c.get("step_roofline"), "cpu_baseline": c["cpu_baseline"],
                              "command": "python bench.py " + " ".join(cmd[2:])}
            except Exception as ex:  # a leg that cannot be measured is reported as such, never silently dropped
                legs[name] = {"error": f"{type(ex).__name__}: {ex}"}

    config4 = None
    if not args.no_config4 and not args.image and alpha and args.level == 1:
        # BASELINE config 4: ONE 16384^2 RGBA raster, its 1369 tiles cut into contiguous ranges over the N ranks (strong scaling).
        # Its >= 6x target at 8 GPUs is a THROUGHPUT figure (rasters per launch x pipeline slots): the latency of one raster is
        # one longest entropy chain per tile whatever N is.
        b4, p4 = (4 if world == 1 else max(2, 8 // world)), 2
        wl4 = "16384x16384 synthetic 'photo' RGBA8, level -1: ONE fixed raster geometry, 1369 tiles cut into contiguous ranges over the ranks"
        if world == 1:
            # N = 1: a CHILD process, like the RGB legs (measured in this process behind the 183 GB headline leg the same leg
            # reads 26 instead of 29.5-30.6 Gpx/s: fragmented device memory)
            env.torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--image", "16384", "--batch", str(b4), "--pipeline", str(p4), "--steps", str(max(6, args.steps // 4)),
                   "--warmup", "2", "--roofline-reps", "2", "--no-legs", "--no-config4", "--no-cpu"]
            if args.probe_run:
                cmd.append("--probe-run")
            try:
                out_c = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
                c = json.loads([ln for ln in out_c.stdout.splitlines() if ln.startswith("{")][-1])
                config4 = {"workload": wl4, "scaling": "strong", "value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"],
                           "rasters_per_launch": c["config"]["batch"], "pipeline_slots": c["config"]["pipeline_slots"], "tiles_per_rank": c["config"]["tiles_per_rank"],
                           "compressed_bytes": c["config"]["compressed_bytes"], "verified": c["verified"], "hbm_in_use_gb": c["config"]["hbm_in_use_gb"],
                           "command": "python bench.py " + " ".join(cmd[2:])}
            except Exception as ex:
                config4 = {"workload": wl4, "error": f"{type(ex).__name__}: {ex}"}
        else:
            r = run_leg(env, 16384, 16384, True, 1, b4, p4, max(4, args.steps // 8), 1, kind=args.kind)
            r.pop("host_raster")
            config4 = {"workload": wl4, "scaling": "strong", "value": round(r["mpx_s"], 1), "unit": "Mpx/s", "ms_per_step": round(r["ms_per_step"], 3),
                       "rasters_per_launch": r["B"], "pipeline_slots": r["P"], "tiles_per_rank": r["tiles_per_rank"],
                       "compressed_bytes": r["compressed_bytes"], "verified": r["verified"], "hbm_in_use_gb": r["hbm_in_use_gb"]}

    if rank == 0:
        r = main_res
        ch = r["ch"]
        from xpng_amd import api as _api
        lib_name = _api.hip_lib()._name
        if (_api.hip_lib().xpnghip_probes_built() != 0) != bool(args.probe_run):
            raise SystemExit("bench.py: the loaded library flavour does not match --probe-run")
        out = {
            "metric": "Mpixels/s encode+decode (bit-exact vs ref)",
            "value": round(r["mpx_s"], 1),
            "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r["ms_per_step"], 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic '{args.kind}' {'RGBA8' if alpha else 'RGB8'}, level -{args.level} ({'FAST' if args.level == 1 else 'SLOW'}), tile encode + decode "
                                   f"(tile-size walk included), rasters (already normalised: normalize_RGBA is part of single_image.store_ms, not of the pipelined step) and blobs resident in HBM; "
                                   f"batch throughput: {B} distinct rasters per launch x {r['P']} pipelined contexts",
                       "batch": B, "pipeline_slots": r["P"], "tiles": r["tiles"], "tiles_per_rank": r["tiles_per_rank"], "share_px": r["my_px"],
                       "parallelism": f"tile-range x{world}" + (f" + file assembly spread over the ranks (image b on rank b % {world}): one packed message per rank pair and step ({'RCCL send/recv over xGMI' if args.backend == 'nccl' else args.backend})" if world > 1 else ""),
                       "compressed_bytes": r["compressed_bytes"], "distinct_rasters_per_launch": B,
                       "hbm_in_use_gb": r["hbm_in_use_gb"], "env": env_seen, "library": os.path.basename(lib_name)},
            "verified": r["verified"],
            "roofline": roofline_obj(r),
            "step_roofline": step_roofline_obj(r),
            "cpu_baseline": cpu,
        }
        # ONE image through the boundary's own entry points, beside the reference doing the same call on this node's CPUs.
        # (`value` above is batch throughput; this is latency.  A single image is a set of serial entropy chains - the
        # alpha stream of the biggest tile is 148 k dependent rANS steps - so the GPU loses this comparison: see DESIGN.md §6.)
        si = {"workload": f"ONE {W}x{H} image of the workload, strictly serial",
              "on_device_encode_ms": round(r["enc_ms"], 3), "on_device_decode_ms": round(r["dec_ms"], 3),
              "on_device_encode_mpx_s": round(r["my_px"] / r["enc_ms"] / 1e3, 1), "on_device_decode_mpx_s": round(r["my_px"] / r["dec_ms"] / 1e3, 1)}
        if "api_enc_ms" in r:
            si.update({"store_ms": round(r["store_ms"], 3), "load_ms": round(r["api_dec_ms"], 3),
                       "store_mpx_s": round(r["my_px"] / r["store_ms"] / 1e3, 1), "load_mpx_s": round(r["my_px"] / r["api_dec_ms"] / 1e3, 1),
                       "store_load": "what xpng_store / xpng_load execute between their clocks, memory to memory on the caller's host buffers, the C calls alone, best of 5: "
                                     "store = xpnghip_image_begin (upload + normalize_RGBA on the device) -> xpnghip_image_encode_T -> xpnghip_image_end "
                                     "(host/xpng_api.c store_on_device; the reference's region libxpng.c:727-760); load = xpnghip_decode_tiles into a freshly malloc()ed raster (libxpng.c:967-985)",
                       "raw_entry_encode_ms": round(r["api_enc_ms"], 3),
                       "raw_entry": "xpnghip_encode_tiles alone on an already normalised raster (the raw entry point; xpng_store does not call it)"})
            if cpu:
                si["reference_cpu_encode_ms"], si["reference_cpu_decode_ms"] = cpu["encode_ms"], cpu["decode_ms"]
                si["api_vs_reference_cpu"] = {"encode": round(cpu["encode_ms"] / r["store_ms"], 3), "decode": round(cpu["decode_ms"] / r["api_dec_ms"], 3),
                                              "note": "store_ms / load_ms against the reference's CPU call for ONE image (same timed regions); ratio > 1 = this library faster, < 1 = slower"}
        out["single_image"] = si
        out["single_image_encode_ms"], out["single_image_decode_ms"] = si["on_device_encode_ms"], si["on_device_decode_ms"]
        out["single_image_encode_mpx_s"], out["single_image_decode_mpx_s"] = si["on_device_encode_mpx_s"], si["on_device_decode_mpx_s"]
        if args.probe_run:
            out["probe_run"] = True
            out["metric"] = "PROBE RUN, not a benchmark: " + out["metric"]
        if legs:
            out["legs"] = legs
        if config4:
            out["config4"] = config4
        print(json.dumps(out), flush=True)
    if world > 1:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
