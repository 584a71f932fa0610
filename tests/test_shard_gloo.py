"""CPU, world_size 2 over gloo: the multi-GPU sharding rule (contiguous tile ranges + gatherv to rank 0) reproduces
the single-process concatenation.  The per-rank blobs come from the oracle here (no GPU in this container); on the
GPU box the same shard.py code moves blobs produced by the HIP codec over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from xpng_amd.shard import band_rows, gather_blobs, tile_ranges
    from xpng_amd.synth import synth_raster
    W, H = 1500, 1200
    tiles = po.tile_table(W, H, 4)
    t0, t1 = tile_ranges(len(tiles), world)[rank]
    y0, y1 = band_rows(tiles, t0, t1)
    # the rank materialises only its band; tile coordinates stay global
    band = synth_raster("photo", W, y1 - y0, True, y0=y0)
    full = np.zeros((H, W, 4), np.uint8)
    full[y0:y1] = band
    blob = b"".join(po.encode_tile(1, full, t) for t in tiles[t0:t1])
    local = torch.from_numpy(np.frombuffer(blob, np.uint8).copy())
    out, lens = gather_blobs(local, len(blob))
    if rank == 0:
        q.put((out.numpy().tobytes(), lens))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process_encode():
    from oracle import pyoracle as po
    from xpng_amd.synth import synth_raster
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, lens = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = po.encode_tiles(1, synth_raster("photo", 1500, 1200, True))
    assert got == whole and sum(lens) == len(whole) and len(lens) == 2


def _worker_batch(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from xpng_amd.shard import gather_blobs_batch, tile_table, weighted_tile_ranges
    from xpng_amd.synth import synth_raster
    W, H, B = 1000, 900, 3
    tiles = tile_table(W, H)
    t0, t1 = weighted_tile_ranges(tiles, world)[rank]
    bufs, lens = [], []
    for b in range(B):
        full = synth_raster("photo", W, H, True, seed=b + 1)
        blob = b"".join(po.encode_tile(1, full, t) for t in tiles[t0:t1])
        bufs.append(torch.from_numpy(np.frombuffer(blob + b"\0" * 32, np.uint8).copy()))
        lens.append(len(blob))
    outs, table = gather_blobs_batch(bufs, lens)
    # the one-message-per-rank form, first discovering the lengths, then with the table known (no length exchange)
    from xpng_amd.shard import gather_blobs_packed, image_from_packs
    packs, table2 = gather_blobs_packed(bufs, lens)
    packs3, table3 = gather_blobs_packed(bufs, lens, table=table2)
    # destinations spread round robin over the ranks: image b is assembled on rank b % world
    from xpng_amd.shard import exchange_blobs_round_robin, image_from_round_robin
    recv, table4 = exchange_blobs_round_robin(bufs, lens)
    recv5, _ = exchange_blobs_round_robin(bufs, lens, table=table4, scratch={})
    for rv in (recv, recv5):
        for b in range(rank, B, world):
            whole = po.encode_tiles(1, synth_raster("photo", W, H, True, seed=b + 1))
            assert image_from_round_robin(rv, table4, b, rank).numpy().tobytes() == whole, (rank, b)
    if rank == 0:
        assert table2 == table and table3 == table and table4 == table
        for pk in (packs, packs3):
            assert [image_from_packs(pk, table, b).numpy().tobytes() for b in range(B)] == [o.numpy().tobytes() for o in outs]
        q.put(([o.numpy().tobytes() for o in outs], table))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batched_gathers(world):
    """gatherv to rank 0 (per image and packed) and the round-robin exchange (image b assembled on rank b % world), 2 and 3 ranks."""
    from oracle import pyoracle as po
    from xpng_amd.synth import synth_raster
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_batch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, table = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for b in range(3):
        assert got[b] == po.encode_tiles(1, synth_raster("photo", 1000, 900, True, seed=b + 1))
    assert len(table) == world and len(table[0]) == 3


def test_host_tile_table_matches_oracle():
    from oracle import pyoracle as po
    from xpng_amd.shard import tile_table
    for (w, h) in [(1, 1), (444, 444), (445, 444), (100, 2000), (2000, 100), (667, 667), (889, 445), (4096, 4096), (16384, 16384),
                   (16384, 8192), (1334, 265), (3799, 1927), (300, 4000)]:
        assert tile_table(w, h) == po.tile_table(w, h, 4), (w, h)


def test_ranges_cover_all_tiles_once():
    from xpng_amd.shard import band_rows, tile_ranges, weighted_tile_ranges
    from oracle import pyoracle as po
    for n, g in [(1369, 8), (81, 8), (3, 8), (1, 2), (666, 4)]:
        r = tile_ranges(n, g)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    tiles = po.tile_table(16384, 16384, 4)
    assert len(tiles) == 1369
    r = weighted_tile_ranges(tiles, 8)
    assert len(r) == 8 and r[0][0] == 0 and r[-1][1] == 1369 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    px = [sum(t[2] * t[3] for t in tiles[a:b]) for a, b in r]
    assert max(px) / min(px) < 1.1
    y0, y1 = band_rows(tiles, *tile_ranges(1369, 8)[3])
    assert 0 < y0 < y1 <= 16384


def _worker_steps(rank, world, port, q):
    """bench.py's per-step exchange, exactly: table known up front, persistent scratch per pipeline slot, 3 steps whose
    blob CONTENT changes while the lengths stay (a stale pack / receive buffer would show)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xpng_amd.shard import exchange_blobs_round_robin, image_from_round_robin
    B = 5
    table = [[100 + 13 * r + 7 * b for b in range(B)] for r in range(world)]     # known (world x B) length table
    lens = table[rank]

    def blob(r, b, step):  # bytes rank r holds for image b at `step`
        return ((np.arange(table[r][b], dtype=np.int64) * (r + 3) + 11 * b + 101 * step) % 251).astype(np.uint8)

    slots = [{}, {}]  # two pipeline slots, each with its own persistent scratch
    ok = True
    for step in range(3):
        scratch = slots[step % 2]
        bufs = [torch.from_numpy(np.concatenate([blob(rank, b, step), np.zeros(16, np.uint8)])) for b in range(B)]
        recv, t2 = exchange_blobs_round_robin(bufs, lens, table=table, scratch=scratch)
        assert t2 is table
        for b in range(rank, B, world):
            want = np.concatenate([blob(r, b, step) for r in range(world)])
            ok = ok and np.array_equal(image_from_round_robin(recv, table, b, rank).numpy(), want)
    if world > 1:
        assert ("rr_pack" in slots[0]) and any(isinstance(k, tuple) and k[0] == "rr_recv" for k in slots[0])  # buffers were kept
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_exchange_path_over_steps_and_slots():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_steps, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True), (1, True)]
