"""Tile-range sharding of one raster over torch.distributed ranks.

Tiles are coded independently (own first pixel, predictor flags and statistics: reference libxpng.c:542-570,
840-862), so a rank encodes a contiguous tile-index range with no data-path collective.  The only exchange
is the final concatenation (reference libxpng.c:764-769): an all-gather of per-rank byte counts, then a
gatherv of the blob bytes to rank 0 as grouped point-to-point sends (RCCL ncclSend/ncclRecv over xGMI on
the GPU box, gloo on CPU for the tests).  No all-reduce, no ring.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


TILE_AREA = 444 * 444


def tile_table(W: int, H: int) -> List[Tuple[int, int, int, int]]:
    """(x, y, w, h) of every tile in row-major order: the reference's compute_props_of_each_tile (libxpng.c:51-83), needed
    on the host to choose a rank's tile range before any device context exists."""
    def split(length, base):
        n, rem = divmod(length, base)
        first, second = base + rem, base
        if rem > base // 2:
            n += 1
            second = first // 2
            first = second + (first & 1)
        return n, first, second
    if W * H <= TILE_AREA:
        return [(0, 0, W, H)]
    if W < 444:
        bw, bh = W, TILE_AREA // W
    elif H < 444:
        bh, bw = H, TILE_AREA // H
    else:
        bw = bh = 444
    nx, w0, w1 = split(W, bw)
    ny, h0, h1 = split(H, bh)
    out = []
    for j in range(ny):
        y = 0 if j == 0 else (h0 if j == 1 else h0 + h1 + (j - 2) * bh)
        th = h0 if j == 0 else (h1 if j == 1 else bh)
        for i in range(nx):
            x = 0 if i == 0 else (w0 if i == 1 else w0 + w1 + (i - 2) * bw)
            tw = w0 if i == 0 else (w1 if i == 1 else bw)
            out.append((x, y, tw, th))
    return out


def tile_ranges(n_tiles: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [floor(k*T/G), floor((k+1)*T/G))."""
    return [(k * n_tiles // world, (k + 1) * n_tiles // world) for k in range(world)]


def weighted_tile_ranges(tiles: Sequence[Tuple[int, int, int, int]], world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges balanced by pixel count (first-row / first-column tiles are up to 2.25x larger): range k ends at
    the first tile where the running pixel count reaches k/world of the total, and every range keeps at least one tile
    while tiles last (ranks beyond the tile count get empty ranges).  The same rule, in the same integer arithmetic, as
    shard_ranges() in csrc/wrappers.hpp (xpnghip_shard_ranges; tests/test_abi.py cross-checks the two)."""
    n = len(tiles)
    d = min(world, n)
    total = sum(t[2] * t[3] for t in tiles)
    out, start, acc, k = [], 0, 0, 1
    for i, t in enumerate(tiles):
        if k >= d:
            break
        acc += t[2] * t[3]
        if acc * d >= total * k or n - (i + 1) == d - k:
            out.append((start, i + 1))
            start = i + 1
            k += 1
    out.append((start, n))
    while len(out) < world:
        out.append((n, n))
    return out


def band_rows(tiles: Sequence[Tuple[int, int, int, int]], t0: int, t1: int) -> Tuple[int, int]:
    """Raster rows [y0, y1) touched by tiles [t0, t1): a rank only needs this band in HBM."""
    if t0 >= t1:
        return 0, 0
    return min(t[1] for t in tiles[t0:t1]), max(t[1] + t[3] for t in tiles[t0:t1])


def gather_blobs(local, local_len: int, group=None, dst: int = 0):
    """gatherv of per-rank blob bytes to rank `dst`.

    local: 1-D uint8 tensor (device for RCCL, CPU for gloo) holding this rank's concatenated tile blobs in its
    first `local_len` bytes.  Returns (tensor_with_all_blobs_in_rank_order or None, list_of_lengths).
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lens_t = torch.zeros(world, dtype=torch.int64, device=local.device)
    mine = torch.tensor([local_len], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(lens_t, mine, group=group) if local.device.type == "cuda" else \
        dist.all_gather(list(lens_t.split(1)), mine, group=group)
    lens = [int(v) for v in lens_t.tolist()]
    if world == 1:
        return local[:local_len], lens
    if rank == dst:
        total = sum(lens)
        out = torch.empty(total, dtype=torch.uint8, device=local.device)
        ops, o = [], 0
        for r in range(world):
            seg = out[o:o + lens[r]]
            if r == dst:
                seg.copy_(local[:local_len])
            elif lens[r]:
                ops.append(dist.P2POp(dist.irecv, seg, r, group))
            o += lens[r]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out, lens
    if local_len:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local[:local_len].contiguous(), dst, group)]):
            req.wait()
    return None, lens


def gather_blobs_batch(locals_, lens_local, group=None, dst: int = 0):
    """gatherv for a batch: every rank holds B blob buffers (one per image) with lens_local[b] valid bytes each.  One
    all-gather of the B lengths per rank, then ONE grouped send/recv moves every rank's B blobs to rank `dst`, which returns
    a list of B tensors (image b = rank-ordered concatenation of its tile ranges) and the (world x B) length table."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = len(locals_)
    dev = locals_[0].device
    mine = torch.tensor(list(lens_local), dtype=torch.int64, device=dev)
    table = torch.zeros(world * B, dtype=torch.int64, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(table, mine, group=group)
    else:
        dist.all_gather(list(table.split(B)), mine, group=group)
    lens = [[int(v) for v in row] for row in table.view(world, B).tolist()]
    if world == 1:
        return [locals_[b][:lens_local[b]] for b in range(B)], lens
    if rank == dst:
        outs, ops = [], []
        for b in range(B):
            out = torch.empty(sum(lens[r][b] for r in range(world)), dtype=torch.uint8, device=dev)
            o = 0
            for r in range(world):
                seg = out[o:o + lens[r][b]]
                if r == dst:
                    seg.copy_(locals_[b][:lens[r][b]])
                elif lens[r][b]:
                    ops.append(dist.P2POp(dist.irecv, seg, r, group))
                o += lens[r][b]
            outs.append(out)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return outs, lens
    ops = [dist.P2POp(dist.isend, locals_[b][:lens_local[b]], dst, group) for b in range(B) if lens_local[b]]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return None, lens


def gather_blobs_packed(locals_, lens_local, table=None, group=None, dst: int = 0, scratch=None):
    """gatherv for a batch with ONE message per rank: a rank packs its B blobs back to back (one device copy kernel) and sends
    the pack to rank `dst`; rank `dst` receives world-1 packs (7 concurrent xGMI links on an 8-GPU node) and returns
    (packs in rank order, table).  `table` is the (world x B) byte-length table; when the caller already knows it (lengths
    are deterministic for a given raster, and an encode that returns lengths has them on the host anyway) no length
    all-gather and no host synchronisation happens here, so the exchange queues behind the encode on the current stream.
    Image b of the file is the concatenation over ranks r of packs[r][off(r,b) : off(r,b) + table[r][b]] with
    off(r,b) = sum(table[r][:b]) - see `image_from_packs`.  `scratch` (a dict the caller keeps, one per pipeline slot) makes
    the pack and receive buffers persistent, so a steady-state step allocates nothing."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = len(locals_)
    dev = locals_[0].device
    if table is None:
        mine = torch.tensor(list(lens_local), dtype=torch.int64, device=dev)
        flat = torch.zeros(world * B, dtype=torch.int64, device=dev)
        if dev.type == "cuda":
            dist.all_gather_into_tensor(flat, mine, group=group)
        else:
            dist.all_gather(list(flat.split(B)), mine, group=group)
        table = [[int(v) for v in row] for row in flat.view(world, B).tolist()]
    if [int(v) for v in table[rank]] != [int(v) for v in lens_local]:
        raise ValueError("length table row of this rank does not match its blob lengths")
    def buffer(key, nbytes):
        if scratch is None:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        t = scratch.get(key)
        if t is None or t.numel() != nbytes or t.device != dev:
            t = scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return t

    pack = buffer("pack", sum(lens_local))
    if pack.numel():
        torch.cat([locals_[b][:lens_local[b]] for b in range(B)], out=pack)
    if world == 1:
        return [pack], table
    if rank == dst:
        packs, ops = [], []
        for r in range(world):
            if r == dst:
                packs.append(pack)
                continue
            buf = buffer(("recv", r), sum(table[r]))
            packs.append(buf)
            if buf.numel():
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return packs, table
    if pack.numel():
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, pack, dst, group)]):
            req.wait()
    return None, table


def image_from_packs(packs, table, b: int):
    """Tile blobs of image b in file order (rank-ordered tile ranges) out of the per-rank packs `gather_blobs_packed` returns."""
    import torch
    segs = []
    for r, pk in enumerate(packs):
        o = sum(table[r][:b])
        segs.append(pk[o:o + table[r][b]])
    return torch.cat(segs)


def exchange_blobs_round_robin(locals_, lens_local, table=None, group=None, scratch=None):
    """The final concatenation with the destinations spread over the ranks: the file of image b is assembled on rank
    b % world.  Every rank packs, per destination, the blobs of that destination's images (one device copy) and the ranks
    exchange ONE message per ordered pair - on an 8-GPU node 56 messages over 56 directed xGMI links, each carrying 1/8 of a
    rank's bytes, instead of 7 messages converging on rank 0 (whose 7 inbound links would carry the whole step: the point-
    to-point fabric has no switch to spread that).  Same byte total, no hot rank.

    Returns (recv, table): recv[r] = the bytes rank r sent here (this rank's own share for r == rank), table = the
    (world x B) length table (learnt by an all-gather when not passed in; with it nothing here synchronises with the host).
    `image_from_round_robin(recv, table, b, rank)` cuts image b (b % world == rank) out of them."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = len(locals_)
    dev = locals_[0].device
    if table is None:
        mine = torch.tensor(list(lens_local), dtype=torch.int64, device=dev)
        flat = torch.zeros(world * B, dtype=torch.int64, device=dev)
        if dev.type == "cuda":
            dist.all_gather_into_tensor(flat, mine, group=group)
        else:
            dist.all_gather(list(flat.split(B)), mine, group=group)
        table = [[int(v) for v in row] for row in flat.view(world, B).tolist()]
    if [int(v) for v in table[rank]] != [int(v) for v in lens_local]:
        raise ValueError("length table row of this rank does not match its blob lengths")

    def buffer(key, nbytes):
        if scratch is None:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        t = scratch.get(key)
        if t is None or t.numel() != nbytes or t.device != dev:
            t = scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return t

    send_sz = [sum(table[rank][b] for b in range(d, B, world)) for d in range(world)]
    recv_sz = [sum(table[r][b] for b in range(rank, B, world)) for r in range(world)]
    pack = buffer("rr_pack", sum(send_sz))
    order = [b for d in range(world) for b in range(d, B, world)]
    if pack.numel():
        torch.cat([locals_[b][:lens_local[b]] for b in order], out=pack)
    send_off = [sum(send_sz[:d]) for d in range(world)]
    recv, ops = [], []
    for r in range(world):
        if r == rank:
            recv.append(pack[send_off[rank]:send_off[rank] + send_sz[rank]])
            continue
        buf = buffer(("rr_recv", r), recv_sz[r])
        recv.append(buf)
        if recv_sz[r]:
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
        if send_sz[r]:
            ops.append(dist.P2POp(dist.isend, pack[send_off[r]:send_off[r] + send_sz[r]], r, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recv, table


def image_from_round_robin(recv, table, b: int, rank: int):
    """Tile blobs of image b (assembled on rank b % world == rank) in file order, out of what exchange_blobs_round_robin returned."""
    import torch
    world = len(table)
    assert b % world == rank
    segs = []
    for r in range(world):
        o = sum(table[r][bb] for bb in range(rank, b, world))
        segs.append(recv[r][o:o + table[r][b]])
    return torch.cat(segs)
