#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle parity report (development aid; the pass/fail gates are tests/ -m gpu).

For every case: chooser sums/pr -> symbol planes -> context streams / k words -> rANS blocks -> tile blobs ->
full concatenation, then decode of oracle-made blobs.  Keeps going after a mismatch and prints where the
first difference is, so one GPU call gives as much information as possible.
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import xpng_amd  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from xpng_amd.api import walk_tile_offsets  # noqa: E402
from xpng_amd.synth import special_cases, synth_raster  # noqa: E402

LOG = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.append(s)


def first_diff(a: np.ndarray, b: np.ndarray):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    if len(d):
        i = int(d[0])
        return f"first diff @{i}: gpu={a[max(0,i-2):i+6].tolist()} ref={b[max(0,i-2):i+6].tolist()} ({len(d)} diffs)"
    return f"length gpu={len(a)} ref={len(b)}"


def check_encode(name, raster, max_tiles_detail=3):
    h, w, ch = raster.shape
    ok_all = True
    ctx = xpng_amd.Context(w, h, ch)
    d_r = torch.from_numpy(raster).cuda()
    d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    t0 = time.time()
    n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
    dt = time.time() - t0
    gpu = d_b[:n].cpu().numpy().tobytes()
    ref = po.encode_tiles(1, raster)
    tiles = ctx.tiles()
    assert tiles == po.tile_table(w, h, ch), "tile table mismatch"
    same = gpu == ref
    say(f"[enc] {name:34s} {w}x{h}x{ch} tiles={len(tiles):4d} bytes gpu={len(gpu)} ref={len(ref)} {'OK' if same else 'MISMATCH'} ({dt*1e3:.1f} ms)")
    if same:
        ctx.close()
        return True
    ok_all = False
    shown = 0
    ro, _ = walk_tile_offsets(ref, len(tiles))
    for ti, t in enumerate(tiles):
        pr_ref, sums_ref = po.choose_predictor(raster, t)
        planes = po.m1_planes(raster, t, pr_ref)
        st = po.m1_streams(raster, t, planes)
        bad = []
        pr = int(ctx.fetch("pr", ti)[0])
        sums = ctx.fetch("sums", ti).view(np.uint32).tolist()
        if pr != pr_ref or sums != sums_ref:
            bad.append(f"pr gpu={pr} ref={pr_ref} sums gpu={sums} ref={sums_ref}")
        for k in ("nl", "r", "g", "b") + (("a",) if ch == 4 else ()):
            g = ctx.fetch(k, ti)
            if not np.array_equal(g, planes[k]):
                bad.append(f"plane {k}: " + first_diff(g, planes[k]))
        for c in range(9):
            g = ctx.fetch(10 + c, ti)
            if not np.array_equal(g, st["ctx"][c]):
                bad.append(f"ctx{c}: " + first_diff(g, st["ctx"][c]))
        g = ctx.fetch("k", ti).view(np.uint32)
        if not np.array_equal(g, st["k"]):
            bad.append("k: " + first_diff(g, st["k"]))
        for c in range(10 if ch == 4 else 9):
            g = ctx.fetch(20 + c, ti).tobytes()
            if c < 9:
                r = po.rans2_encode(st["F"][c], 9, st["ctx"][c], 12)
            else:
                r = po.rans2_encode(st["FA"], 256, planes["a"][1:], 15)
            if g != r:
                bad.append(f"blk{c} (n={len(st['ctx'][c]) if c < 9 else len(planes['a'])-1}): gpu {len(g)}B hdr={g[:12].hex()} ref {len(r)}B hdr={r[:12].hex()} "
                           + first_diff(np.frombuffer(g, np.uint8), np.frombuffer(r, np.uint8)))
        if bad:
            shown += 1
            say(f"   tile {ti} {t}:")
            for b in bad[:12]:
                say("      " + b)
            if shown >= max_tiles_detail:
                break
    if shown == 0:
        say("   all stages equal per tile; container/concatenation differs: " +
            first_diff(np.frombuffer(gpu, np.uint8), np.frombuffer(ref, np.uint8)))
    ctx.close()
    return ok_all


def parse_m2_blob(blob):
    """Split a mode-2 colour tile blob into (type, b-bytes, [17 blocks])."""
    h0 = int.from_bytes(blob[:4], "little")
    ty = h0 >> 24
    if ty in (0, 255) or (ty >> 4) == 2:
        return ty, blob[4:], []
    bsz = int.from_bytes(blob[4:8], "little")
    o = 4 + bsz
    blocks = []
    for _ in range(17):
        sz = int.from_bytes(blob[o:o + 4], "little") & 0xFFFFFF
        blocks.append(blob[o:o + sz]); o += sz
    return ty, blob[4:4 + bsz], blocks


def check_encode_m2(name, raster):
    h, w, ch = raster.shape
    if ch != 3:
        return True
    ctx = xpng_amd.Context(w, h, ch)
    d_r = torch.from_numpy(raster).cuda()
    d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(2, d_r.data_ptr(), d_b.data_ptr())
    gpu = d_b[:n].cpu().numpy().tobytes()
    ref = po.encode_tiles(2, raster)
    same = gpu == ref
    say(f"[enc2] {name:33s} {w}x{h}x{ch} tiles={ctx.n_tiles:4d} bytes gpu={len(gpu)} ref={len(ref)} {'OK' if same else 'MISMATCH'}")
    if not same:
        ro, _ = walk_tile_offsets(ref, ctx.n_tiles)
        try:
            go, _ = walk_tile_offsets(gpu, ctx.n_tiles)
        except Exception:
            go = ro
        for ti in range(ctx.n_tiles):
            rb = ref[ro[ti]:ro[ti + 1] if ti + 1 < ctx.n_tiles else len(ref)]
            gb = gpu[go[ti]:go[ti + 1] if ti + 1 < ctx.n_tiles else len(gpu)]
            if rb == gb:
                continue
            say(f"   tile {ti} {ctx.tile(ti)}: gpu hdr={gb[:8].hex()} ({len(gb)}B) ref hdr={rb[:8].hex()} ({len(rb)}B)")
            try:
                gt, gbits, gblk = parse_m2_blob(gb); rt, rbits, rblk = parse_m2_blob(rb)
                if gbits != rbits:
                    say("      b differs: " + first_diff(np.frombuffer(gbits, np.uint8), np.frombuffer(rbits, np.uint8)))
                for k, (x, y) in enumerate(zip(gblk, rblk)):
                    if x != y:
                        say(f"      block {k}: gpu {len(x)}B {x[:12].hex()} ref {len(y)}B {y[:12].hex()} " + first_diff(np.frombuffer(x, np.uint8), np.frombuffer(y, np.uint8)))
            except Exception as e:
                say(f"      (parse failed: {e!r})")
            break
    ctx.close()
    return same


def check_decode(name, raster):
    h, w, ch = raster.shape
    blobs = po.encode_tiles(1, raster)
    ctx = xpng_amd.Context(w, h, ch)
    off, total = walk_tile_offsets(blobs, ctx.n_tiles)
    d_b = torch.from_numpy(np.frombuffer(blobs + b"\0" * 64, dtype=np.uint8).copy()).cuda()
    d_r = torch.zeros(h * w * ch + 64, dtype=torch.uint8, device="cuda")
    t0 = time.time()
    ctx.decode_device(1, d_b.data_ptr(), len(blobs), off, d_r.data_ptr())
    torch.cuda.synchronize()
    dt = time.time() - t0
    out = d_r[: h * w * ch].cpu().numpy().reshape(h, w, ch)
    same = np.array_equal(out, raster)
    say(f"[dec] {name:34s} {w}x{h}x{ch} tiles={ctx.n_tiles:4d} {'OK' if same else 'MISMATCH'} ({dt*1e3:.1f} ms)")
    if not same:
        d = np.argwhere(out != raster)
        y, x, c = d[0]
        say(f"   {len(d)} byte diffs; first at y={y} x={x} c={c}: gpu={out[y, x].tolist()} ref={raster[y, x].tolist()}; "
            f"rows with diffs: {np.unique(d[:,0])[:8].tolist()} cols: {np.unique(d[:,1])[:8].tolist()}")
        tiles = ctx.tiles()
        for ti, t in enumerate(tiles):
            sub_o = out[t[1]:t[1] + t[3], t[0]:t[0] + t[2]]
            sub_r = raster[t[1]:t[1] + t[3], t[0]:t[0] + t[2]]
            if not np.array_equal(sub_o, sub_r):
                dd = np.argwhere(sub_o != sub_r)
                say(f"   tile {ti} {t} type=0x{blobs[off[ti]+3]:02x}: {len(dd)} diffs, first (y,x,c)={dd[0].tolist()} chans={np.unique(dd[:,2]).tolist()}")
                break
    ctx.close()
    return same


def check_decode_m2(name, raster):
    h, w, ch = raster.shape
    if ch != 3:
        return True
    blobs = po.encode_tiles(2, raster)
    ctx = xpng_amd.Context(w, h, ch)
    off, total = walk_tile_offsets(blobs, ctx.n_tiles)
    d_b = torch.from_numpy(np.frombuffer(blobs + b"\0" * 64, dtype=np.uint8).copy()).cuda()
    d_r = torch.zeros(h * w * ch + 64, dtype=torch.uint8, device="cuda")
    ctx.decode_device(2, d_b.data_ptr(), len(blobs), off, d_r.data_ptr())
    torch.cuda.synchronize()
    out = d_r[: h * w * ch].cpu().numpy().reshape(h, w, ch)
    same = np.array_equal(out, raster)
    say(f"[dec2] {name:33s} {w}x{h}x{ch} tiles={ctx.n_tiles:4d} {'OK' if same else 'MISMATCH'}")
    if not same:
        d = np.argwhere(out != raster)
        for ti, t in enumerate(ctx.tiles()):
            so, sr = out[t[1]:t[1] + t[3], t[0]:t[0] + t[2]], raster[t[1]:t[1] + t[3], t[0]:t[0] + t[2]]
            if not np.array_equal(so, sr):
                dd = np.argwhere(so != sr)
                say(f"   tile {ti} {t} type=0x{blobs[off[ti]+3]:02x}: {len(dd)} diffs, first (y,x,c)={dd[0].tolist()} gpu={so[tuple(dd[0][:2])].tolist()} ref={sr[tuple(dd[0][:2])].tolist()}")
                break
    ctx.close()
    return same


def cases(level):
    yield "photo_64x64_rgba", synth_raster("photo", 64, 64, True)
    yield "photo_64x64_rgb", synth_raster("photo", 64, 64, False)
    yield "photo_5x7_rgb", synth_raster("photo", 5, 7, False)
    yield "photo_4x4_rgba", synth_raster("photo", 4, 4, True)
    yield "photo_1x1_rgb", synth_raster("photo", 1, 1, False)
    yield "photo_2x1_rgb", synth_raster("photo", 2, 1, False)
    yield "flat_64x64_rgb", synth_raster("flat", 64, 64, False)
    yield "flat_64x64_rgba", synth_raster("flat", 64, 64, True)
    yield "noise_64x64_rgb", synth_raster("noise", 64, 64, False)
    yield "noise_64x64_rgba", synth_raster("noise", 64, 64, True)
    yield "gray_445x444_rgb", synth_raster("gray", 445, 444, False)
    yield "photo_444x444_rgba", synth_raster("photo", 444, 444, True)
    yield "photo_700x500_rgb", synth_raster("photo", 700, 500, False)
    yield "photo_700x500_rgba", synth_raster("photo", 700, 500, True)
    yield "noise_700x500_rgba", synth_raster("noise", 700, 500, True)
    yield "photo_100x2000_rgba", synth_raster("photo", 100, 2000, True)
    yield "photo_2000x100_rgb", synth_raster("photo", 2000, 100, False)
    if level >= 1:
        yield "photo_667x667_rgba", synth_raster("photo", 667, 667, True)
        yield "photo_1500x1200_rgba", synth_raster("photo", 1500, 1200, True)
        yield "photo_1500x1200_rgb", synth_raster("photo", 1500, 1200, False)
        for n, r in special_cases():
            r = po.normalize_rgba(r)
            yield "special_" + n, r
        gold = os.path.join(ROOT, "tests", "golden")
        from xpng_amd.synth import load_seven
        for f in sorted(os.listdir(gold)):
            if f.endswith(".7"):
                yield f[:-2], load_seven(os.path.join(gold, f))
    if level >= 2:
        yield "photo_4096_rgba", synth_raster("photo", 4096, 4096, True)
        yield "photo_4096_rgb", synth_raster("photo", 4096, 4096, False)


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    what = sys.argv[2] if len(sys.argv) > 2 else "both"
    say("devices:", xpng_amd.device_count(), torch.cuda.get_device_name(0))
    bad = 0
    for name, r in cases(level):
        r = np.ascontiguousarray(r)
        for fn in ({"enc": [check_encode], "dec": [check_decode], "enc2": [check_encode_m2], "dec2": [check_decode_m2],
                    "m2": [check_encode_m2, check_decode_m2]}.get(what, [check_encode, check_decode])):
            try:
                if not fn(name, r):
                    bad += 1
            except Exception as e:
                bad += 1
                say(f"[EXC] {fn.__name__} {name}: {e!r}")
                say(traceback.format_exc())
    say("TOTAL FAILURES:", bad)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "check.log"), "w") as f:
        f.write("\n".join(LOG) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
