#!/usr/bin/env python3
"""bench.py -- xPNG hot path on MI355X: Mpixels/s encode+decode, bit-exact vs the reference.

  python bench.py [--gpus N --steps K --warmup W]            (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one synthetic raster already resident in HBM:
    level-1 tile ENCODE (raster -> concatenated tile blobs)  [+ RCCL gatherv of blobs to rank 0 when N>1]
  + level-1 tile DECODE (blobs -> raster).
Workload: each rank owns a 4096x4096-pixel share of a synthetic `photo` RGBA raster; at N=1 that
is BASELINE.json's "4096x4096 synthetic RGBA8, level -1" configuration, at N ranks the global raster is
(4096*a)x(4096*b), a*b=N, cut into contiguous tile ranges (weak scaling; tiles are independent, reference
libxpng.c:542-570).  `--image 16384` instead fixes the global raster at 16384^2 (strong scaling, config 4).

Before timing, the output is verified: md5(header + blobs) against the reference-generated manifest when the
workload is pinned there, and decode(encode(x)) == x always.

One JSON line on rank 0.  `roofline` is the bandwidth-bound kernel pair of the path, predictor chooser +
per-pixel transform (BASELINE config 2), timed live with HIP events on its stream; `cpu_baseline` is the compiled
reference (oracle/_ref/xpng, kind "reference") or the oracle's C port timed on this node's host cores.
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


def grid_for(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2), 16: (4, 4)}.get(n, (n, 1))


def cpu_baseline(raster_np, budget_s=25.0):
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
                e = subprocess.run([ref_bin, "-1", src, dst], capture_output=True, text=True)
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
                    "sample": f"compiled reference libxpng.c (build.sh flags), xpng -1 / -d on the same {w}x{h}x{ch} raster, best of {runs} runs, T=min(tiles,nproc)"}
    from oracle import pyoracle as po  # CPU port as the fallback baseline
    best_e = best_d = 0.0
    for _ in range(3):
        data = po.encode_image(1, raster_np)
        best_e = max(best_e, px / (po.last_encode_ns() / 1e9) / 1e6)
        po.decode_image(data)
        best_d = max(best_d, px / (po.last_decode_ns() / 1e9) / 1e6)
    return {"value": round(1.0 / (1.0 / best_e + 1.0 / best_d), 1), "unit": "Mpx/s", "cores": min(cores, 81), "kind": "port",
            "encode_mpx_s": round(best_e, 1), "decode_mpx_s": round(best_d, 1), "host_cpus": cores,
            "sample": f"oracle C port, encode+decode of the same {w}x{h}x{ch} raster, best of 3"}


def ctx_len_at(ctx, i):
    from xpng_amd.api import hip_lib
    return hip_lib().xpnghip_ctx_last_blobs_len_at(ctx._h, i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--image", type=int, default=0, help="fix the GLOBAL raster at image x image pixels (strong scaling)")
    ap.add_argument("--share", type=int, default=4096, help="per-rank share edge in pixels (weak scaling)")
    ap.add_argument("--rgb", action="store_true", help="RGB instead of RGBA")
    ap.add_argument("--kind", default="photo")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--roofline-reps", type=int, default=50)
    ap.add_argument("--pipeline", type=int, default=5,
                    help="contexts (each with its own HIP stream and buffers) that consecutive steps alternate between, so the "
                         "encode of step k+1 runs beside the decode of step k; 1 = strictly serial steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo moves blobs through the host, for rehearsals)")
    ap.add_argument("--batch", type=int, default=64,
                    help="images per launch per rank: a step encodes+decodes `batch` rasters of the workload in one batched launch "
                         "sequence (the entropy stage is a serial chain per tile stream, so one image alone cannot fill 256 CUs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import xpng_amd
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.shard import band_rows, exchange_blobs_round_robin, image_from_round_robin, tile_table, weighted_tile_ranges
    from xpng_amd.synth import seven_header, synth_raster_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the xPNG tile codec has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % ndev  # (a rehearsal may run several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    alpha = not args.rgb
    ch = 4 if alpha else 3
    if args.image:
        W = H = args.image
        scaling = "strong"
    else:
        a, b = grid_for(world)
        W, H = args.share * a, args.share * b
        scaling = "weak"
    B = max(1, args.batch)
    tiles = tile_table(W, H)
    t0, t1 = weighted_tile_ranges(tiles, world)[rank]
    # one context, B images per launch, workspace only for this rank's tile range
    ctx = xpng_amd.Context(W, H, ch, device=local_rank, batch=B, tile_range=(t0, t1))
    assert ctx.tiles() == tiles
    y0, y1 = band_rows(tiles, t0, t1)
    # the rank materialises only the raster band its tiles touch (+ one spare row: the staged 16-byte loads of the last
    # row may run a few bytes past it when the band is not the tail of the image).  The B rasters of a launch are DISTINCT
    # (seeds 1..B; image 0 is the one the manifest pins): identical rasters would sit in the 256 MB Infinity Cache for the
    # roofline kernels and would make every lane of the wide entropy kernels take the same branches.
    dev = f"cuda:{local_rank}"
    band_stores = [synth_raster_torch(args.kind, W, min(H, y1 + 1) - y0, alpha, seed=1 + b, y0=y0, device=dev) for b in range(B)]
    bands = [bs[: y1 - y0] for bs in band_stores]
    band = bands[0]
    bpr = W * ch
    d_raster_virtual = band.data_ptr() - y0 * bpr  # kernels address rows absolutely; only [y0, y1) is ever touched
    d_blobs_all = [torch.empty(ctx.blob_bound(t0, t1) + 64, dtype=torch.uint8, device=band.device) for _ in range(B)]
    d_back_all = [torch.zeros_like(band) for _ in range(B)]
    d_blobs, d_back = d_blobs_all[0], d_back_all[0]
    d_back_virtual = d_back.data_ptr() - y0 * bpr
    stream = torch.cuda.current_stream().cuda_stream
    rast_ptrs = [bd.data_ptr() - y0 * bpr for bd in bands]
    blob_ptrs = [t.data_ptr() for t in d_blobs_all]
    back_ptrs = [t.data_ptr() - y0 * bpr for t in d_back_all]
    # pipeline slots: slot 0 is (ctx, current stream, the buffers above); further slots get their own context / stream / buffers
    P = max(1, args.pipeline)
    # (slot 0 gets a stream of its own for the steps as well: the default stream is the legacy NULL stream)
    slots = [dict(ctx=ctx, stream=torch.cuda.Stream(), blobs=d_blobs_all, back=d_back_all, blob_ptrs=blob_ptrs, back_ptrs=back_ptrs)]
    for _ in range(P - 1):
        bl = [torch.empty_like(d_blobs_all[0]) for _ in range(B)]
        bk = [torch.zeros_like(band) for _ in range(B)]
        slots.append(dict(ctx=xpng_amd.Context(W, H, ch, device=local_rank, batch=B, tile_range=(t0, t1)), stream=torch.cuda.Stream(),
                          blobs=bl, back=bk, blob_ptrs=[t.data_ptr() for t in bl], back_ptrs=[t.data_ptr() - y0 * bpr for t in bk]))
    step_no = [0]
    my_px = sum(t[2] * t[3] for t in tiles[t0:t1])
    total_px = W * H

    def same_tiles(a, b):  # rows shared with a neighbouring rank's tiles are not written by this rank: compare tile by tile
        if world == 1:
            return bool(torch.equal(a, b))
        return all(bool(torch.equal(a[ty - y0:ty - y0 + th, tx:tx + tw], b[ty - y0:ty - y0 + th, tx:tx + tw])) for (tx, ty, tw, th) in tiles[t0:t1])

    # ---- correctness gate (untimed): bit-exact vs reference manifest where pinned (image 0), round trip for every image
    n = ctx.encode_device(1, d_raster_virtual, d_blobs.data_ptr(), t0, t1, stream=stream)
    blob0 = d_blobs[:n].clone()
    lens_b = ctx.encode_device_batch(1, rast_ptrs, blob_ptrs, t0, t1, stream=stream)   # all B images, one launch sequence
    ok = lens_b[0] == n and bool(torch.equal(d_blobs_all[0][:n], blob0))                # batched == single-image launch
    offs_b = []
    for bi in range(B):
        o, tot = walk_tile_offsets(d_blobs_all[bi][:lens_b[bi]].cpu().numpy().tobytes(), t1 - t0)
        ok = ok and tot == lens_b[bi]
        offs_b.append(o)
    off = offs_b[0]
    ctx.decode_device_batch(1, blob_ptrs, lens_b, offs_b, back_ptrs, t0, t1, stream=stream)
    torch.cuda.synchronize()
    ok = ok and ctx.decode_status() == 0
    for bi in range(B):
        ok = ok and same_tiles(d_back_all[bi], bands[bi])
    verified = {"roundtrip": bool(ok), "images_verified": B}
    use_host = world > 1 and args.backend != "nccl"

    len_table = [None]

    def exchange(bufs=None, scratch=None):
        # the one exchange of the path: the file of image b is assembled on rank b % world, so every rank sends each other
        # rank ONE message per step (its tile-range blobs of that rank's images, packed): 56 messages over 56 directed xGMI
        # links on a node instead of 7 converging on rank 0.  RCCL send/recv, no collective on the data path.  The first call
        # learns the (world x B) length table; later calls pass it in, so nothing here synchronises with the host and the
        # exchange queues behind the encode on its stream
        bufs = d_blobs_all if bufs is None else bufs
        if use_host:
            bufs = [t[:lens_b[i]].cpu() for i, t in enumerate(bufs)]
        recv, table = exchange_blobs_round_robin(bufs, lens_b, table=len_table[0], scratch=scratch)
        len_table[0] = table
        return recv, table

    if world > 1:
        recv, table = exchange()
        gathered, lens = (image_from_round_robin(recv, table, 0, 0) if rank == 0 else None), [row[0] for row in table]
    else:
        gathered, lens = d_blobs_all[0][:n], [n]
    if rank == 0:
        man_path = os.path.join(ROOT, "tests", "golden", "manifest.json")
        key = f"synth_{args.kind}_{W}x{H}_{'rgba' if alpha else 'rgb'}"
        if os.path.exists(man_path):
            man = json.load(open(man_path))
            if key in man and "L1" in man[key]:
                md = hashlib.md5(seven_header(W, H, alpha, level=1) + gathered.cpu().numpy().tobytes()).hexdigest()
                verified["reference_size"] = 8 + int(sum(lens)) == man[key]["L1"]["size"]
                verified["reference_md5"] = md == man[key]["L1"]["md5"]
                ok = ok and verified["reference_md5"]
    if not ok:
        raise SystemExit(f"rank {rank}: output is NOT bit-exact / does not round-trip: {verified}")
    ref_blobs = [d_blobs_all[bi][:lens_b[bi]].clone() for bi in range(B)]  # what every slot must reproduce

    def step():
        # one launch sequence covers all B images (virtual tile = image * N + tile); consecutive steps alternate pipeline slots
        sl = slots[step_no[0] % P]
        step_no[0] += 1
        sh = sl["stream"].cuda_stream
        sl["ctx"].encode_device_batch(1, rast_ptrs, sl["blob_ptrs"], t0, t1, stream=sh, sync=False)
        if world > 1:
            with torch.cuda.stream(sl["stream"]):
                exchange(sl["blobs"], sl.setdefault("xchg", {}))  # tile bytes are deterministic: the lengths are the ones verified above
        sl["ctx"].decode_device_batch(1, sl["blob_ptrs"], lens_b, offs_b, sl["back_ptrs"], t0, t1, stream=sh)

    # ---- per-stage rates on this rank (HIP events on the stream the kernels run on)
    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps  # ms

    # the roofline kernels run over the whole batch per launch (a single 4096^2 pass is ~30 us, i.e. launch-bound); timed
    # here, before the pipelined steps, with nothing else on the device
    tr_ms = timed(lambda: ctx.transform_device_batch(rast_ptrs, t0, t1, stream=stream), args.roofline_reps)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every pipeline slot allocates its decode workspace on first use: touch each once before the W warmup steps, so that a
    # small W still leaves no allocation inside the timed region (these P untimed steps are in addition to the W requested)
    for _ in range(P):
        step()
    barrier()
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=band.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # every image of every pipeline slot must have produced the verified bytes and raster
    for si, sl in enumerate(slots):
        for bi in range(B):
            if ctx_len_at(sl["ctx"], bi) != lens_b[bi] or not torch.equal(sl["blobs"][bi][:lens_b[bi]], ref_blobs[bi]) or not same_tiles(sl["back"][bi], bands[bi]):
                raise SystemExit(f"rank {rank}: slot {si} image {bi} differs from the verified image")

    enc_ms = timed(lambda: ctx.encode_device(1, d_raster_virtual, d_blobs.data_ptr(), t0, t1, stream=stream, sync=False), max(3, args.steps // 2))
    dec_ms = timed(lambda: ctx.decode_device(1, d_blobs.data_ptr(), n, off, d_back_virtual, t0, t1, stream=stream), max(3, args.steps // 2))
    algo_bytes = ALGO_BYTES_PER_PX[ch] * my_px * B
    achieved = algo_bytes / (tr_ms * 1e-3) / 1e9

    if rank == 0:
        # HBM bytes per launch of the roofline kernels from the committed PMC passes (FETCH_SIZE doubled + WRITE_SIZE, per pixel):
        # counters cannot be collected inside this run, so the figure is the profile's bytes/px times this launch's pixels
        traffic, traffic_src = None, None
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_transform.json")
        if alpha and args.kind == "photo" and os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                traffic = int(pmc["per_pixel_bytes"]["total"] * B * my_px)
                traffic_src = "profiles/r01_pmc_transform.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/px of that run x pixels of this launch)"
            except Exception:
                traffic, traffic_src = None, None
        out = {
            "metric": "Mpixels/s encode+decode (bit-exact vs ref)",
            "value": round(B * total_px * args.steps / elapsed / 1e6, 1),
            "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic '{args.kind}' {'RGBA8' if alpha else 'RGB8'}, level -1 (FAST), tile encode + decode, rasters and blobs resident in HBM",
                       "batch": B, "pipeline_slots": P, "tiles": len(tiles), "tiles_per_rank": t1 - t0, "share_px": my_px,
                       "parallelism": f"tile-range x{world}" + (f" + file assembly spread over the ranks (image b on rank b % {world}): one packed message per rank pair and step ({'RCCL send/recv over xGMI' if args.backend == 'nccl' else args.backend})" if world > 1 else ""),
                       "compressed_bytes": int(sum(lens)), "distinct_rasters_per_launch": B,
                       "hbm_in_use_gb": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30, 1)},
            "verified": verified,
            "single_image_encode_mpx_s": round(my_px / enc_ms / 1e3, 1), "single_image_decode_mpx_s": round(my_px / dec_ms / 1e3, 1),
            "single_image_encode_ms": round(enc_ms, 3), "single_image_decode_ms": round(dec_ms, 3),
            "roofline": {"kernel": "k_chooser + k_m1_transform (predictor chooser + per-pixel transform, BASELINE config 2)",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_px": ALGO_BYTES_PER_PX[ch], "ms_per_launch": round(tr_ms, 4),
                         "transform_mpx_s": round(B * my_px / tr_ms / 1e3, 1), "images_per_launch": B,
                         "read_only_frac_of_peak": round(ch * B * my_px / (tr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(band.cpu().numpy())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl["ctx"].close()


if __name__ == "__main__":
    main()
