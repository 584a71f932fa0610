"""One rank of tests/test_rccl_exchange.py (started as a child process; never collected by pytest).
argv: rank world port result_dir.  Each rank owns one GPU, encodes ITS contiguous tile range of every image of a small batch with
the HIP codec, takes part in the one exchange of the path (xpng_amd/shard.py exchange_blobs_round_robin over RCCL: the
concatenation of libxpng.c:764-769, image b assembled on rank b % world) and checks the images assembled here against the
oracle's tile bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import xpng_amd
    from oracle import pyoracle as po  # checker only
    from xpng_amd.shard import band_rows, exchange_blobs_round_robin, image_from_round_robin, tile_table, weighted_tile_ranges
    from xpng_amd.synth import synth_raster
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    checked = 0
    for (W, H, alpha, level) in [(1501, 1203, True, 1), (1501, 1203, False, 2)]:
        ch, B = (4 if alpha else 3), 3
        tiles = tile_table(W, H)
        t0, t1 = weighted_tile_ranges(tiles, world)[rank]
        y0, y1 = band_rows(tiles, t0, t1)
        ctx = xpng_amd.Context(W, H, ch, device=rank, batch=B, tile_range=(t0, t1))
        rasters = [synth_raster("photo", W, H, alpha, seed=11 + b) for b in range(B)]
        bpr = W * ch
        # the rank holds only the band its tiles touch (+ one spare row, as bench.py does); kernels address rows absolutely
        bands = [torch.from_numpy(r[y0:min(H, y1 + 1)].copy()).to(dev) for r in rasters]
        blobs = [torch.empty(ctx.blob_bound(t0, t1) + 64, dtype=torch.uint8, device=dev) for _ in range(B)]
        lens = ctx.encode_device_batch(level, [bd.data_ptr() - y0 * bpr for bd in bands], [t.data_ptr() for t in blobs], t0, t1,
                                       stream=torch.cuda.current_stream().cuda_stream)
        recv, table = exchange_blobs_round_robin(blobs, lens)
        recv2, _ = exchange_blobs_round_robin(blobs, lens, table=table, scratch={})   # the steady-state form: table known, no host sync
        for b in range(rank, B, world):
            want = po.encode_image(level, rasters[b])[8:]
            for rv in (recv, recv2):
                got = image_from_round_robin(rv, table, b, rank).cpu().numpy().tobytes()
                assert got == want, (rank, W, H, alpha, level, b, len(got), len(want))
                checked += 1
        ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(outdir, f"ok_{rank}"), "w") as f:
        f.write(str(checked))


if __name__ == "__main__":
    main()
