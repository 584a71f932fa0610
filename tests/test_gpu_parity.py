"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI (libxpng_hip.so /
libxpng.so via ctypes); the oracle and the reference-generated goldens are only the checker."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, corpus_entries, corpus_raster, golden_raster, small_entries

pytestmark = pytest.mark.gpu


def md5(b):
    return hashlib.md5(b).hexdigest()


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xpng_amd
    if not torch.cuda.is_available() or xpng_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    return xpng_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def test_native_library_is_the_one_loaded(gpu):
    from xpng_amd import api
    assert os.path.samefile(api.hip_lib()._name, api.HIP_SO)
    maps = open("/proc/self/maps").read()
    assert "xpng_amd/lib/libxpng_hip.so" in maps


def test_store_matches_reference_goldens_all_levels(gpu, manifest, po, tmp_path):
    """xpng_store (host C driver -> GPU tile codec) reproduces the reference's .xpng bytes for every small golden, at
    levels 1 (FAST), 2 (SLOW: RGB through the 17-stream rANS v1 coder, gray and single-colour tiles; RGBA falls back to
    level 1 as libxpng.c:755 does) and 7."""
    checked = 0
    for name, ent in small_entries(manifest):
        raster = golden_raster(name, ent)
        for level in (1, 2, 7):
            g = ent.get(f"L{level}")
            if g is None:
                continue
            out = tmp_path / "o.xpng"
            gpu.store(level, raster, str(out))
            data = out.read_bytes()
            assert len(data) == g["size"] and md5(data) == g["md5"], (name, level)
            if "file" in g:
                assert data == open(os.path.join(GOLD, g["file"]), "rb").read()
            checked += 1
    assert checked >= 400


def test_load_decodes_reference_goldens(gpu, manifest, po, tmp_path):
    """xpng_load on files PRODUCED BY THE REFERENCE (stored goldens, else oracle bytes pinned to the golden md5)."""
    from xpng_amd.synth import to_seven_bytes
    checked = 0
    for name, ent in small_entries(manifest):
        raster = golden_raster(name, ent)
        for level in (1, 2, 7):
            g = ent.get(f"L{level}")
            if g is None:
                continue
            if "file" in g:
                data = open(os.path.join(GOLD, g["file"]), "rb").read()
            else:
                data = po.encode_image(level, raster)
                assert md5(data) == g["md5"]
            p = tmp_path / "i.xpng"
            p.write_bytes(data)
            back = gpu.load(str(p))
            assert md5(to_seven_bytes(back)) == g["decoded_md5"], (name, level)
            checked += 1
    assert checked >= 400


@pytest.mark.parametrize("kind,w,h,alpha", [("photo", 700, 500, True), ("photo", 1500, 1200, False), ("noise", 700, 500, True),
                                            ("photo", 100, 2000, True), ("photo", 2000, 100, False), ("flat", 889, 445, True),
                                            ("photo", 1333, 901, False), ("noise", 1501, 1203, False), ("photo", 447, 446, False)])
def test_stage_planes_and_streams_match_oracle(gpu, po, kind, w, h, alpha):
    """BASELINE config 2 parity: chooser + per-pixel transform planes, then streams and rANS blocks, per tile."""
    import torch
    from xpng_amd.synth import synth_raster
    raster = synth_raster(kind, w, h, alpha)
    ch = raster.shape[2]
    ctx = gpu.Context(w, h, ch)
    d_r = torch.from_numpy(raster).cuda()
    d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    assert ctx.tiles() == po.tile_table(w, h, ch)
    # BASELINE config-2 entry point: chooser + transform, the five symbol planes left in HBM
    ctx.transform_device(d_r.data_ptr())
    all_planes = []
    for ti, t in enumerate(ctx.tiles()):
        pr, sums = po.choose_predictor(raster, t)
        assert ctx.fetch("sums", ti).view(np.uint32).tolist() == sums
        assert int(ctx.fetch("pr", ti)[0]) == pr
        planes = po.m1_planes(raster, t, pr)
        for k in ("nl", "r", "g", "b") + (("a",) if ch == 4 else ()):
            assert np.array_equal(ctx.fetch(k, ti), planes[k]), (ti, k)
        all_planes.append((pr, planes))
    # full encode, then the rANS blocks
    ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
    for ti, t in enumerate(ctx.tiles()):
        pr, planes = all_planes[ti]
        assert int(ctx.fetch("pr", ti)[0]) == pr
        if ch == 4:
            assert np.array_equal(ctx.fetch("a", ti)[1:], planes["a"][1:]), ti   # the alpha symbol plane the rANS block read
        st = po.m1_streams(raster, t, planes)
        for c in range(9):
            assert np.array_equal(ctx.fetch(10 + c, ti), st["ctx"][c]), (ti, c)
        assert np.array_equal(ctx.fetch("k", ti).view(np.uint32), st["k"])
        for c in range(9):
            assert ctx.fetch(20 + c, ti).tobytes() == po.rans2_encode(st["F"][c], 9, st["ctx"][c], 12)
        if ch == 4:
            assert ctx.fetch(29, ti).tobytes() == po.rans2_encode(st["FA"], 256, planes["a"][1:], 15)
    ctx.close()


def test_transform_only_entry_point(gpu, po):
    import torch
    from xpng_amd.synth import synth_raster
    raster = synth_raster("photo", 1000, 900, True)
    ctx = gpu.Context(1000, 900, 4)
    d_r = torch.from_numpy(raster).cuda()
    ctx.transform_device(d_r.data_ptr())
    for ti, t in enumerate(ctx.tiles()):
        pr, _ = po.choose_predictor(raster, t)
        planes = po.m1_planes(raster, t, pr)
        for k in ("nl", "r", "g", "b", "a"):
            assert np.array_equal(ctx.fetch(k, ti), planes[k]), (ti, k)
    ctx.close()


@pytest.mark.parametrize("name,level", [("synth_photo_4096x4096_rgba", 1), ("synth_photo_4096x4096_rgb", 1), ("synth_photo_4096x4096_rgb", 2),
                                        ("synth_noise_4096x4096_rgba", 1), ("synth_noise_4096x4096_rgb", 2)])
def test_full_size_4096_matches_reference_md5(gpu, manifest, name, level, tmp_path):
    """BASELINE config 3 at full size (photo RGBA at -1, photo RGB at -1 and -2, noise = raw tiles / level-7 rewrite): file md5
    equals what the compiled reference wrote (manifest), and the decode of that file returns the source raster."""
    ent = manifest[name]
    raster = golden_raster(name, ent)
    out = tmp_path / "big.xpng"
    gpu.store(level, raster, str(out))
    data = out.read_bytes()
    assert len(data) == ent[f"L{level}"]["size"] and md5(data) == ent[f"L{level}"]["md5"]
    back = gpu.load(str(out))
    assert np.array_equal(back, raster)


def test_pipeline_shards_of_the_host_buffer_calls(gpu, manifest, tmp_path):
    """An ordinary large image goes through xpnghip_encode_tiles / xpnghip_decode_tiles as three tile-row groups on ONE device
    (wrappers.hpp pipeline_shards: transfers overlap the chains; the first tile row first, its band the last to come back through
    pinned staging and copy threads).  Bytes are those of the reference; a corrupt or truncated body is rejected on that path
    too (a wrong size word in any group), and the raster handed in is complete when the call returns 0."""
    from xpng_amd import api
    name = "synth_photo_4096x4096_rgba"
    ent = manifest[name]
    raster = golden_raster(name, ent)
    blobs = api.encode_tiles(1, raster)
    from xpng_amd.synth import seven_header
    assert md5(seven_header(4096, 4096, True, level=1) + blobs) == ent["L1"]["md5"]
    out = np.empty_like(raster)                                  # untouched pages, like xpng_load's malloc
    api.decode_tiles(1, blobs, 4096, 4096, 4, out=out)
    assert np.array_equal(out, raster)
    off, total = api.walk_tile_offsets(blobs, 81)
    assert total == len(blobs)
    for t in (0, 8, 9, 40, 80):                                   # a tile of every group, first and last
        bad = bytearray(blobs)
        bad[off[t] + 4:off[t] + 8] = b"\xff\xff\xff\x7f"          # k size beyond the tile's blob
        with pytest.raises(api.XpngError):
            api.decode_tiles(1, bytes(bad), 4096, 4096, 4)
    with pytest.raises(api.XpngError):
        api.decode_tiles(1, blobs[: len(blobs) // 2], 4096, 4096, 4)
    out2 = np.empty_like(raster)
    api.decode_tiles(1, blobs, 4096, 4096, 4, out=out2)          # and the path is intact afterwards
    assert np.array_equal(out2, raster)
    # RGB, level 2, odd geometry: bands start at every 16-byte phase
    from xpng_amd.synth import synth_raster
    from oracle import pyoracle as po
    r2 = synth_raster("photo", 3001, 2999, False, seed=9)
    want = po.encode_tiles(2, r2)
    assert api.encode_tiles(2, r2) == want
    assert np.array_equal(api.decode_tiles(2, want, 3001, 2999, 3), r2)


def test_tile_range_sharding_concatenates_to_whole(gpu, po):
    """Tiles are independent: encoding [0,k) and [k,N) separately and concatenating equals the whole (multi-GPU rule)."""
    import torch
    from xpng_amd.synth import synth_raster
    raster = synth_raster("photo", 1500, 1200, True)
    ctx = gpu.Context(1500, 1200, 4)
    d_r = torch.from_numpy(raster).cuda()
    d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    whole_n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
    whole = d_b[:whole_n].cpu().numpy().tobytes()
    parts = b""
    for a, b in ((0, 4), (4, 5), (5, 9)):
        n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr(), t0=a, t1=b)
        parts += d_b[:n].cpu().numpy().tobytes()
    assert parts == whole == po.encode_tiles(1, raster)
    ctx.close()


def test_cli_roundtrip_like_reference_test_rb(gpu, manifest, tmp_path):
    """reference test.rb:28-38: xpng -o src.7 res.xpng && xpng -d res.xpng res.7 && cmp src.7 res.7, o in {1,2,7}."""
    from xpng_amd import api
    for name in ("img_pigz-logo", "crop_2021", "img_juicy"):
        src = os.path.join(GOLD, name + ".7")
        for o in ("1", "2", "7"):
            res, back = tmp_path / "res.xpng", tmp_path / "res.7"
            r = subprocess.run([api.CLI, "-" + o, src, str(res)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            if o != "7":
                assert "encode," in r.stdout and "MPx/s" in r.stdout
                assert md5(res.read_bytes()) == manifest[name]["L" + o]["md5"]
            r = subprocess.run([api.CLI, "-d", str(res), str(back)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert back.read_bytes() == open(src, "rb").read()


def test_16384_photo_rgba_matches_reference_md5_and_roundtrips(gpu, manifest):
    """BASELINE config 4 geometry on one GPU: 16384^2 synthetic RGBA generated in HBM, encoded, md5 of header+blobs
    equals the compiled reference's output (manifest), decode returns the source."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import seven_header, synth_raster_torch
    ent = manifest.get("synth_photo_16384x16384_rgba")
    W = H = 16384
    d_r = synth_raster_torch("photo", W, H, True)
    ctx = gpu.Context(W, H, 4)
    assert ctx.n_tiles == 1369
    d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
    blobs = d_b[:n].cpu().numpy().tobytes()
    if ent is not None:
        assert 8 + n == ent["L1"]["size"]
        assert md5(seven_header(W, H, True, level=1) + blobs) == ent["L1"]["md5"]
    off, total = walk_tile_offsets(blobs, ctx.n_tiles)
    assert total == n
    d_out = torch.zeros(W * H * 4 + 64, dtype=torch.uint8, device="cuda")
    ctx.decode_device(1, d_b.data_ptr(), n, off, d_out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_out[: W * H * 4].view(H, W, 4), d_r)
    ctx.close()


def test_batched_launch_equals_single_image_launches(gpu, po):
    """One batched launch over B different rasters produces, per image, exactly the bytes of a single-image launch."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import synth_raster
    W, H, B = 900, 700, 3
    rs = [synth_raster("photo", W, H, True, seed=s + 1) for s in range(B)]
    ctx = gpu.Context(W, H, 4, batch=B)
    d_r = [torch.from_numpy(r).cuda() for r in rs]
    d_b = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    lens = ctx.encode_device_batch(1, [t.data_ptr() for t in d_r], [t.data_ptr() for t in d_b])
    blobs = [d_b[i][:lens[i]].cpu().numpy().tobytes() for i in range(B)]
    for i in range(B):
        assert blobs[i] == po.encode_tiles(1, rs[i]), i
    offs = [walk_tile_offsets(b, ctx.n_tiles)[0] for b in blobs]
    d_o = [torch.zeros(W * H * 4 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    ctx.decode_device_batch(1, [t.data_ptr() for t in d_b], lens, offs, [t.data_ptr() for t in d_o])
    torch.cuda.synchronize()
    for i in range(B):
        assert np.array_equal(d_o[i][: W * H * 4].cpu().numpy().reshape(H, W, 4), rs[i]), i
    ctx.close()


def test_mode2_device_entry_points_match_oracle(gpu, po):
    """Mode 2 (RGB) through the device-resident C-ABI: colour, gray, single-colour and raw tiles in one raster."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import special_cases
    for name, raster in special_cases():
        raster = np.ascontiguousarray(po.normalize_rgba(raster))
        if raster.shape[2] != 3:
            continue
        h, w, _ = raster.shape
        ctx = gpu.Context(w, h, 3)
        d_r = torch.from_numpy(raster).cuda()
        d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        n = ctx.encode_device(2, d_r.data_ptr(), d_b.data_ptr())
        blobs = d_b[:n].cpu().numpy().tobytes()
        assert blobs == po.encode_tiles(2, raster), name
        off, _ = walk_tile_offsets(blobs, ctx.n_tiles)
        d_o = torch.zeros(h * w * 3 + 64, dtype=torch.uint8, device="cuda")
        ctx.decode_device(2, d_b.data_ptr(), n, off, d_o.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_o[: h * w * 3].cpu().numpy().reshape(h, w, 3), raster), name
        ctx.close()


def test_corrupt_files_are_rejected_not_executed(gpu, po, tmp_path):
    """The reference decoder trusts the file (libxpng.c:982: no bounds checks).  On the GPU a wild offset is a device fault, so
    every tile header is validated first: corrupt or truncated files make xpng_load fail cleanly, and good data still decodes."""
    from xpng_amd.synth import synth_raster
    rng = np.random.default_rng(5)
    for alpha, level in ((True, 1), (False, 1), (False, 2)):
        raster = synth_raster("photo", 700, 500, alpha)
        good = po.encode_image(level, raster)
        p = tmp_path / "x.xpng"
        rejected = 0
        for trial in range(24):
            bad = bytearray(good)
            if trial % 3 == 0:                      # truncate
                bad = bad[: 8 + int(rng.integers(1, len(good) - 8))]
            elif trial % 3 == 1:                    # smash a header-ish word near the start of a tile
                o = 8 + int(rng.integers(0, 64))
                bad[o:o + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            else:                                   # random words anywhere
                for _ in range(8):
                    o = int(rng.integers(8, len(bad) - 4))
                    bad[o:o + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            p.write_bytes(bytes(bad))
            try:
                out = gpu.load(str(p))
                assert out.shape == raster.shape   # accepted: payload corruption only changes pixel values, never faults
            except gpu.XpngError:
                rejected += 1
        assert rejected >= 8
        p.write_bytes(good)
        assert np.array_equal(gpu.load(str(p)), raster)


def test_batch_kernels_forced_on_small_inputs(gpu, manifest, po, tmp_path, monkeypatch):
    """Large batches switch the entropy stage to its "wide" kernels (many streams per wavefront: rans2_wide.hpp,
    rans2_wide_dec.hpp, k_dec_walk_wide) and to 256-thread workgroups.  XPNG_WIDE_RANS=1 forces that path for any input, so
    the goldens, the edge cases and the corrupt-file behaviour are checked on it too."""
    monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    checked = 0
    for name, ent in small_entries(manifest):
        raster = golden_raster(name, ent)
        want = np.ascontiguousarray(po.normalize_rgba(raster))  # (what the file holds: hidden colours zeroed, opaque alpha dropped)
        for level in (1, 2):   # level 2: the wide rANS v1 kernels (rans1_wide.hpp, rans1_wide_dec.hpp)
            g = ent.get(f"L{level}")
            if g is None:
                continue
            out = tmp_path / "o.xpng"
            gpu.store(level, raster, str(out))
            data = out.read_bytes()
            assert len(data) == g["size"] and md5(data) == g["md5"], (name, level)
            back = gpu.load(str(out))
            assert back.shape == want.shape and np.array_equal(back, want), (name, level)
            checked += 1
    assert checked >= 200
    from xpng_amd.synth import synth_raster
    rng = np.random.default_rng(11)
    raster = synth_raster("photo", 700, 500, True)
    good = po.encode_image(1, raster)
    p = tmp_path / "x.xpng"
    for trial in range(12):
        bad = bytearray(good)
        for _ in range(8):
            o = int(rng.integers(8, len(bad) - 4))
            bad[o:o + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        p.write_bytes(bytes(bad))
        try:
            assert gpu.load(str(p)).shape == raster.shape
        except gpu.XpngError:
            pass
    p.write_bytes(good)
    assert np.array_equal(gpu.load(str(p)), raster)


@pytest.mark.parametrize("alpha,split", [(True, False), (True, True), (False, False), (False, True)])
def test_large_batch_takes_the_wide_path_and_matches(gpu, po, monkeypatch, alpha, split):
    """A batch big enough to select the wide kernels by itself (tiles x streams > 2048); RGBA and RGB (the one-wave-per-tile
    band reconstruction writes 16-byte and 12-byte pixel groups respectively); with and without the decode tail split by tile
    size class (the default since round 4: the big-tile class on the side stream behind the alpha chains, the small-tile class on
    the caller's stream; XPNG_NO_SPLIT=1: one walk, one tail)."""
    if not split:
        monkeypatch.setenv("XPNG_NO_SPLIT", "1")
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import synth_raster
    W, H, B = 1800, 1500, 18 if alpha else 21   # 12 tiles x 10 (9) streams x B images > 2048 chains
    ch = 4 if alpha else 3
    base = [synth_raster(k, W, H, alpha, seed=s + 1) for s, k in enumerate(("photo", "noise", "photo"))]
    ctx = gpu.Context(W, H, ch, batch=B)
    assert ctx.n_tiles * (10 if alpha else 9) * B > 2048
    d_r = [torch.from_numpy(base[i % 3]).cuda() for i in range(B)]
    d_b = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    lens = ctx.encode_device_batch(1, [t.data_ptr() for t in d_r], [t.data_ptr() for t in d_b])
    want = [po.encode_tiles(1, r) for r in base]
    offs = []
    for i in range(B):
        blob = d_b[i][:lens[i]].cpu().numpy().tobytes()
        assert blob == want[i % 3], i
        offs.append(walk_tile_offsets(blob, ctx.n_tiles)[0])
    d_o = [torch.full((W * H * ch + 64,), 0xA5, dtype=torch.uint8, device="cuda") for _ in range(B)]
    ctx.decode_device_batch(1, [t.data_ptr() for t in d_b], lens, offs, [t.data_ptr() for t in d_o])
    torch.cuda.synchronize()
    assert ctx.decode_status() == 0
    for i in range(B):
        assert np.array_equal(d_o[i][: W * H * ch].cpu().numpy().reshape(H, W, ch), base[i % 3]), i
        assert bool((d_o[i][W * H * ch:] == 0xA5).all()), i   # nothing written behind the raster
    ctx.close()


def test_wide_alpha_chains_on_adversarial_alpha_planes(gpu, po):
    """The wide alpha ENCODE chain (rans2_wide.hpp) gathers its table entries from tables interleaved across the 32 streams of a
    wavefront, two entry sets used alternately [r4]: alpha planes that stress it - constant (one symbol: no chain at all), two values,
    all 256 values, a single odd pixel, ramps - mixed inside ONE launch, so that streams of every kind share wavefronts; 37 rasters of
    6 tiles (222 alpha streams: six full groups of 32 and a ragged seventh).  Bytes against the oracle, then the decode round trip."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import synth_raster
    W, H, B = 1210, 700, 37
    rng = np.random.default_rng(11)
    base = []
    for kind in range(6):
        r = synth_raster("photo" if kind % 2 else "noise", W, H, True, seed=20 + kind).copy()
        if kind == 0: r[..., 3] = 200                                        # one symbol everywhere
        elif kind == 1: r[..., 3] = np.where(rng.random((H, W)) < 0.02, 3, 250)  # two values, one rare
        elif kind == 2: r[..., 3] = rng.integers(1, 256, (H, W))              # every value, no structure
        elif kind == 3: r[..., 3] = 255; r[H // 2, W // 3, 3] = 17            # opaque but for one pixel
        elif kind == 4: r[..., 3] = (np.arange(W)[None, :] // 5 + np.arange(H)[:, None] // 3) % 255 + 1   # ramps: a few small deltas
        else: r[..., 3] = np.where((np.arange(W)[None, :] // 64 + np.arange(H)[:, None] // 64) % 2 == 0, 1, 255)  # blocks: rare big jumps
        base.append(np.ascontiguousarray(r))
    ctx = gpu.Context(W, H, 4, batch=B)
    assert ctx.n_tiles * 10 * B > 2048 and (ctx.n_tiles * B) % 32 != 0
    d_r = [torch.from_numpy(base[i % 6]).cuda() for i in range(B)]
    d_b = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    want = [po.encode_tiles(1, r) for r in base]
    for rep in range(2):  # (twice: the second launch finds the first one's tables in the interleaved region)
        lens = ctx.encode_device_batch(1, [t.data_ptr() for t in d_r], [t.data_ptr() for t in d_b])
        offs = []
        for i in range(B):
            blob = d_b[i][:lens[i]].cpu().numpy().tobytes()
            assert blob == want[i % 6], (rep, i)
            offs.append(walk_tile_offsets(blob, ctx.n_tiles)[0])
    d_o = [torch.full((W * H * 4 + 64,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(B)]
    ctx.decode_device_batch(1, [t.data_ptr() for t in d_b], lens, offs, [t.data_ptr() for t in d_o])
    torch.cuda.synchronize()
    assert ctx.decode_status() == 0
    for i in range(B):
        assert np.array_equal(d_o[i][: W * H * 4].cpu().numpy().reshape(H, W, 4), base[i % 6]), i
    ctx.close()


def test_normalize_rgba_on_device_matches_oracle(gpu, po):
    """normalize_RGBA (libxpng.c:688-721) as device kernels: hidden colour -> zeroed, opaque -> RGB, translucent -> unchanged;
    odd pixel counts exercise the tails of the 4-pixel-per-thread kernels."""
    import torch
    from xpng_amd.synth import synth_raster
    rng = np.random.default_rng(3)
    for (w, h), case in [((301, 203), "hidden"), ((301, 203), "opaque"), ((300, 200), "translucent"), ((7, 5), "hidden"), ((5, 3), "opaque")]:
        r = synth_raster("noise", w, h, True)
        if case == "opaque":
            r[..., 3] = 255
        elif case == "translucent":
            r[..., 3] = np.maximum(r[..., 3], 1)
            r[0, 0, 3] = 200
        else:
            r[..., 3] = rng.integers(0, 3, (h, w)) * 127
        want = np.ascontiguousarray(po.normalize_rgba(r))
        d_in = torch.from_numpy(r).cuda()
        d_out = torch.zeros(w * h * 4 + 64, dtype=torch.uint8, device="cuda")
        pxsz, rewritten = gpu.normalize_device(d_in.data_ptr(), w * h, d_out.data_ptr())
        assert pxsz == want.shape[2], case
        got = d_out[: w * h * pxsz].cpu().numpy().reshape(h, w, pxsz) if rewritten else r
        assert np.array_equal(got, want), case
        assert rewritten == (case != "translucent"), case


def test_orientation_search_sizes_match_oracle(gpu, po, tmp_path, monkeypatch, capsys):
    """tools/orient_search.py (the reference's Mirroring_and_Rotating/test.rb on the GPU: 8 orientations, batched per
    geometry): every listed size equals what the oracle's encode_image gives for that orientation."""
    import importlib.util
    import sys
    from xpng_amd.synth import synth_raster, to_seven_bytes
    spec = importlib.util.spec_from_file_location("orient_search", os.path.join(os.path.dirname(GOLD), "..", "tools", "orient_search.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for alpha in (False, True):
        r = synth_raster("photo", 610, 470, alpha)
        p = tmp_path / "o.7"
        p.write_bytes(to_seven_bytes(r))
        monkeypatch.setattr(sys, "argv", ["orient_search.py", str(p), "2"])
        sizes = mod.main()
        assert len(sizes) == 8
        want = {}
        cur = r
        for rot in ("   0", "  90", " 180", " 270"):
            ms = ("    ", " + v", " + h") if rot in ("   0", "  90") else ("    ",)
            for m in ms:
                v = cur if m == "    " else (cur[::-1] if m == " + v" else cur[:, ::-1])
                want[rot + m] = len(po.encode_image(2, np.ascontiguousarray(v)))
            cur = np.rot90(cur, k=-1)
        assert sizes == want


@pytest.mark.parametrize("force_wide", [False, True])
def test_extreme_aspect_ratios(gpu, po, monkeypatch, force_wide):
    """Tiles far from 444 x 444 (libxpng.c:51-83 keeps w*h near 444^2: a 4-pixel-high raster gets tiles ~49 000 px wide, beyond
    the LDS-staged transform and the band reconstruction, so the generic kernels are used): encode == oracle, decode round-trips,
    on the narrow and on the wide entropy path."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import synth_raster
    if force_wide:
        monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    for (w, h, a) in [(30000, 4, True), (4, 30000, True), (50000, 5, False), (7, 9000, False), (100000, 4, True)]:
        r = synth_raster("photo", w, h, a)
        ch = r.shape[2]
        ctx = gpu.Context(w, h, ch)
        d_r = torch.from_numpy(r).cuda()
        d_b = torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
        blob = d_b[:n].cpu().numpy().tobytes()
        assert blob == po.encode_tiles(1, r), (w, h, a)
        off, _ = walk_tile_offsets(blob, ctx.n_tiles)
        d_o = torch.zeros(w * h * ch + 64, dtype=torch.uint8, device="cuda")
        ctx.decode_device(1, d_b.data_ptr(), n, off, d_o.data_ptr())
        torch.cuda.synchronize()
        assert ctx.decode_status() == 0
        assert np.array_equal(d_o[: w * h * ch].cpu().numpy().reshape(h, w, ch), r), (w, h, a)
        ctx.close()


def test_mode2_large_batch_takes_the_wide_path_and_matches(gpu, po):
    """Level 2 (RGB) in a batch big enough to select the wide rANS v1 kernels by itself; colour, gray and single-colour tiles."""
    import torch
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import special_cases, synth_raster
    mixed = dict(special_cases())["mixed_tiles"]                      # 1000 x 900: colour, gray and single-colour tiles
    h, w, _ = mixed.shape
    base = [mixed, synth_raster("photo", w, h, False, seed=3), synth_raster("noise", w, h, False, seed=5)]
    B = 33
    ctx = gpu.Context(w, h, 3, batch=B)
    assert ctx.n_tiles * 17 * B > 2048
    d_r = [torch.from_numpy(np.ascontiguousarray(base[i % 3])).cuda() for i in range(B)]
    d_b = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    lens = ctx.encode_device_batch(2, [t.data_ptr() for t in d_r], [t.data_ptr() for t in d_b])
    want = [po.encode_tiles(2, np.ascontiguousarray(r)) for r in base]
    offs = []
    for i in range(B):
        blob = d_b[i][:lens[i]].cpu().numpy().tobytes()
        assert blob == want[i % 3], i
        offs.append(walk_tile_offsets(blob, ctx.n_tiles)[0])
    d_o = [torch.zeros(w * h * 3 + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
    ctx.decode_device_batch(2, [t.data_ptr() for t in d_b], lens, offs, [t.data_ptr() for t in d_o])
    torch.cuda.synchronize()
    assert ctx.decode_status() == 0
    for i in range(B):
        assert np.array_equal(d_o[i][: w * h * 3].cpu().numpy().reshape(h, w, 3), base[i % 3]), i
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k_syms", [3, 30, 47, 49, 64, 200])
def test_alpha_alphabet_sizes_on_the_wide_path(gpu, po, monkeypatch, k_syms):
    """The wide alpha encode chain gathers its table entries from the stream's 4 KB table in global memory one block ahead (a
    compact LDS form for alphabets of at most 48 symbols existed through round 3); the wide alpha decode chain uses 16-bit
    cumulative tables with a coarse table indexed by cold rank.  Alphabets from 3 to 200 symbols, bytes against the oracle, then
    the round trip."""
    monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    from xpng_amd.synth import synth_raster
    rng = np.random.default_rng(k_syms)
    W, H = 600, 520
    raster = synth_raster("photo", W, H, True, seed=5).copy()
    steps = rng.choice(np.arange(1, 256), size=k_syms - 1, replace=False)          # alpha deltas in use (plus 0)
    d = np.where(rng.random((H, W)) < 0.7, 0, rng.choice(steps, size=(H, W))).astype(np.int64)
    a = (np.cumsum(d, axis=1) + 1) & 255
    raster[..., 3] = a.astype(np.uint8)
    raster[raster[..., 3] == 0] = 0
    want = po.encode_image(1, raster)
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "a.xpng")
        gpu.store(1, raster, p)
        data = open(p, "rb").read()
        assert data == want, k_syms
        back = gpu.load(p)
    assert np.array_equal(back, np.ascontiguousarray(po.normalize_rgba(raster)))


@pytest.mark.gpu
def test_misaligned_tile_blobs_on_the_wide_path(gpu, po, monkeypatch, tmp_path):
    """RGB with odd tile sizes: raw tiles (w*h*3 + 4 bytes, not a multiple of 4) in front of coded tiles, so the k words
    and rANS blocks of the coded tiles start at every byte alignment (the wide residual kernel cuts its bit window out of
    aligned dwords; the wide chains load words through unaligned-safe paths)."""
    monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    from xpng_amd.synth import synth_raster
    W, H = 1799, 1499
    raster = synth_raster("photo", W, H, False, seed=3).copy()
    noise = synth_raster("noise", W, H, False, seed=4)
    raster[:, : W // 2] = noise[:, : W // 2]        # left tile columns become raw tiles, the right ones stay coded
    raster[H // 2:, :] = np.where((np.arange(W) // 300 % 2 == 0)[None, :, None], noise[H // 2:], raster[H // 2:])
    want = po.encode_image(1, raster)
    p = tmp_path / "m.xpng"
    gpu.store(1, raster, str(p))
    assert p.read_bytes() == want
    assert np.array_equal(gpu.load(str(p)), raster)
    from xpng_amd.api import walk_tile_offsets
    offs, _ = walk_tile_offsets(want[8:], 12)
    assert any(o % 4 and want[8 + o + 3] != 0 for o in offs)   # coded tiles at odd offsets: the case is what it claims to be


@pytest.mark.parametrize("force_wide", [False, True])
def test_whole_reference_corpus_both_directions(gpu, manifest, tmp_path, monkeypatch, force_wide):
    """BASELINE config 5 / reference test.rb:28-38 on its own image set, whole images (multi-tile photographs, the three RGBA
    images, the 1-bit flat one): xpng_load of every reference-written .xpng returns the reference's .7, and xpng_store at
    levels 1 and 2 writes the reference's bytes.  17 images x 2 levels x 2 directions, on the narrow (one wave per chain) and
    the forced wide (many chains per wave) entropy kernels."""
    from xpng_amd.synth import to_seven_bytes
    if force_wide:
        monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    done = 0
    for name, ent in corpus_entries(manifest):
        raster = corpus_raster(ent)            # (oracle decode of the golden, pinned to seven_md5: checker only)
        for level in (1, 2):
            g = ent[f"L{level}"]
            gold_path = os.path.join(GOLD, g["file"])
            back = gpu.load(gold_path)                                   # decode direction: the reference's file
            assert md5(to_seven_bytes(back)) == g["decoded_md5"], (name, level, "decode")
            out = tmp_path / "o.xpng"
            gpu.store(level, raster, str(out))                           # encode direction: the reference's bytes
            assert out.read_bytes() == open(gold_path, "rb").read(), (name, level, "encode")
            done += 1
    assert done == 34


def test_device_side_size_walk(gpu, po):
    """xpnghip_decode_device[_batch] with tile_off == NULL: the serial size walk of libxpng.c:982 runs on the device (one lane per
    image), for a caller whose blobs never leave HBM.  Same pixels as with host-walked offsets; a zeroed size word parks the
    rest of that image's tiles and the decode reports it instead of faulting."""
    import torch
    from xpng_amd.synth import synth_raster
    for (W, H, alpha, level) in [(1500, 1200, True, 1), (1501, 1203, False, 1), (1501, 1203, False, 2)]:
        ch = 4 if alpha else 3
        B = 3
        rs = [synth_raster("photo", W, H, alpha, seed=s + 1) for s in range(B)]
        ctx = gpu.Context(W, H, ch, batch=B)
        d_r = [torch.from_numpy(r).cuda() for r in rs]
        d_b = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in range(B)]
        lens = ctx.encode_device_batch(level, [t.data_ptr() for t in d_r], [t.data_ptr() for t in d_b])
        d_o = [torch.zeros(W * H * ch, dtype=torch.uint8, device="cuda") for _ in range(B)]
        ctx.decode_device_batch(level, [t.data_ptr() for t in d_b], lens, None, [t.data_ptr() for t in d_o])
        assert ctx.decode_status() == 0
        for i in range(B):
            assert np.array_equal(d_o[i].cpu().numpy().reshape(H, W, ch), rs[i]), (W, H, alpha, level, i)
        # single-image entry point, then a broken size chain in image 1
        d_o[0].zero_()
        ctx.decode_device(level, d_b[0].data_ptr(), lens[0], None, d_o[0].data_ptr())
        assert ctx.decode_status() == 0 and np.array_equal(d_o[0].cpu().numpy().reshape(H, W, ch), rs[0])
        first = int.from_bytes(d_b[1][:4].cpu().numpy().tobytes(), "little") & 0xFFFFFF
        d_b[1][first:first + 3] = 0                                  # size field of tile 1 := 0
        ctx.decode_device_batch(level, [t.data_ptr() for t in d_b], lens, None, [t.data_ptr() for t in d_o])
        assert ctx.decode_status() == 1
        ctx.close()


def test_level2_stream_counts_that_do_not_fit_together_are_rejected(gpu, po):
    """Level-2 decode: the 17 decoded streams of a tile lie back to back in a region sized for what a real tile can hold (n - 1
    context symbols, 3 (n - 1) class symbols), each in the room its block header's symbol count asks for.  A file whose class
    blocks EACH claim the per-stream maximum passes the per-stream bound the format gives and must still be rejected - before any
    chain writes a byte - and the untouched file must decode afterwards."""
    import torch
    from xpng_amd.synth import synth_raster
    W, H = 420, 300                                   # one tile
    raster = synth_raster("photo", W, H, False, seed=9)
    ctx = gpu.Context(W, H, 3)
    assert ctx.n_tiles == 1
    n = W * H
    good = po.encode_tiles(2, raster)
    u32 = lambda b, o: int.from_bytes(b[o:o + 4], "little")
    assert (u32(good, 0) >> 28) == 1                  # a colour tile (libxpng.c:929-961)
    bsz = u32(good, 4)
    o, heads = 4 + bsz, []
    for slot in range(17):                            # the 17 block headers behind the shared bit stream
        hdr = u32(good, o)
        heads.append((slot, o, hdr >> 24, hdr & 0xFFFFFF))
        o += hdr & 0xFFFFFF
    assert o == (u32(good, 0) & 0xFFFFFF)
    bad = bytearray(good)
    grown = 0
    for slot, off, typ, size in heads:
        if slot >= 11 and typ in (3, 4):              # class >= 3 streams: 3 n symbols is the most the format allows one of them
            bad[off + 4:off + 8] = (3 * n).to_bytes(4, "little")
            grown += 1
    assert grown >= 2
    d_o = torch.zeros(n * 3 + 64, dtype=torch.uint8, device="cuda")
    for force_wide in (False, True):
        if force_wide:
            os.environ["XPNG_WIDE_RANS"] = "1"
        try:
            d_b = torch.zeros(len(bad) + 64, dtype=torch.uint8, device="cuda")
            d_b[: len(bad)] = torch.frombuffer(bytearray(bad), dtype=torch.uint8).cuda()
            ctx.decode_device(2, d_b.data_ptr(), len(bad), [0], d_o.data_ptr())
            assert ctx.decode_status() == 1
            d_b[: len(good)] = torch.frombuffer(bytearray(good), dtype=torch.uint8).cuda()
            ctx.decode_device(2, d_b.data_ptr(), len(good), [0], d_o.data_ptr())
            assert ctx.decode_status() == 0
            assert np.array_equal(d_o[: n * 3].cpu().numpy().reshape(H, W, 3), raster)
        finally:
            os.environ.pop("XPNG_WIDE_RANS", None)
    ctx.close()


@pytest.mark.parametrize("T", [2, 3])
def test_worker_count_T_shards_tile_ranges_over_devices(gpu, po, tmp_path, monkeypatch, T):
    """xpng_store_T / xpng_load_T with T > 1 (reference libxpng.c:146-151: T workers over the tile cursor): T devices of one
    process each code a contiguous pixel-weighted tile range from their own band of the raster; blob ranges are gathered on
    the first device (peer copies) for the concatenation.  On a one-GPU box the T shards are rehearsed on one device: that
    switch (XPNG_FAKE_DEVICES) exists only in the PROBES build of the library, which this test loads beside the release one.
    The bytes must not depend on T.  Odd widths: band starts at every 16-byte phase."""
    from xpng_amd import api
    from xpng_amd.synth import synth_raster
    P = api.probes_lib()
    monkeypatch.setenv("XPNG_FAKE_DEVICES", str(T))
    assert P.xpnghip_devices_for(T, 1501, 1203) == T
    assert P.xpnghip_devices_for(0, 1501, 1203) == 1          # automatic = ONE device (multi-device is opt-in)
    assert P.xpnghip_devices_for(0, 16384, 16384) == 1
    assert api.hip_lib().xpnghip_devices_for(T, 1501, 1203) == min(T, api.device_count())   # the release library ignores the switch
    for (W, H, alpha, level) in [(1500, 1200, True, 1), (1501, 1203, False, 1), (1501, 1203, False, 2), (1499, 1300, True, 1)]:
        raster = synth_raster("photo", W, H, alpha, seed=7)
        want = po.encode_image(level, raster)
        p = tmp_path / "t.xpng"
        gpu.store(level, raster, str(p), T=T)                  # release library: min(T, real devices) devices
        assert p.read_bytes() == want, (W, H, alpha, level)
        assert np.array_equal(gpu.load(str(p), T=T), raster), (W, H, alpha, level)
        blobs = api.encode_tiles(level, raster, T=T, lib=P)    # T shards (host raster -> bands)
        assert blobs == want[8:]
        assert np.array_equal(api.decode_tiles(level, blobs, W, H, raster.shape[2], T=T, lib=P), raster)
        pxsz, single, blobs2, norm = api.staged_encode(level, raster, T=T, lib=P)   # T shards from the staged device raster (xpng_store's path)
        assert (pxsz, single, blobs2) == (raster.shape[2], False, want[8:]) and norm == raster.tobytes()
    # corrupt file on the multi-device path: rejected, not executed
    bad = bytearray(want[8:])
    bad[4:8] = b"\xff\xff\xff\x7f"                                  # k size of tile 0 beyond its blob
    with pytest.raises(api.XpngError):
        api.decode_tiles(level, bytes(bad), W, H, raster.shape[2], T=T, lib=P)


def test_two_real_devices_code_the_same_bytes(gpu, po):
    """Hardware-gated: with two or more REAL devices xpnghip_encode_tiles_T(T=2) / decode must give the oracle's
    bytes through peer copies.  Skipped on one-GPU boxes - which is every box this suite has run on so far."""
    from xpng_amd import api
    from xpng_amd.synth import synth_raster
    if api.device_count() < 2:
        pytest.skip("one visible device: the multi-device C path has still never run on real peers")
    for (W, H, alpha, level) in [(1501, 1203, True, 1), (1501, 1203, False, 2)]:
        raster = synth_raster("photo", W, H, alpha, seed=3)
        want = po.encode_image(level, raster)[8:]
        assert api.encode_tiles(level, raster, T=2) == want
        assert np.array_equal(api.decode_tiles(level, want, W, H, raster.shape[2], T=2), raster)


def test_concurrent_store_and_load_from_two_host_threads(gpu, manifest, tmp_path):
    """The reference is re-entrant (no globals; every call spawns its own workers: until_fork/4_letters.c:9-17, SURVEY 8(b)
    "Threading").  Two host threads call xpng_store / xpng_load at the same time on different images (ctypes releases the
    GIL): every output must equal the reference's golden bytes, and nothing may dead-lock or serialise on a lock held across
    calls (the staged image is a per-call handle, contexts are checked out of a pool)."""
    import threading
    from xpng_amd.synth import to_seven_bytes
    ents = [(n, e) for n, e in small_entries(manifest) if "L1" in e and "L2" in e][:14]
    assert len(ents) >= 8
    jobs = [(n, e, golden_raster(n, e)) for n, e in ents]
    errors = []

    def worker(k):
        try:
            order = jobs if k == 0 else jobs[::-1]             # both threads visit every image, from opposite ends
            for rep in range(2):
                for i, (n, e, raster) in enumerate(order):
                    for level in (1, 2):
                        g = e.get(f"L{level}")
                        if g is None:
                            continue
                        out = tmp_path / f"t{k}_{i}_{level}.xpng"
                        gpu.store(level, raster, str(out))
                        data = out.read_bytes()
                        if md5(data) != g["md5"]:
                            errors.append((k, n, level, "store"))
                        back = gpu.load(str(out))
                        if md5(to_seven_bytes(back)) != g["decoded_md5"]:
                            errors.append((k, n, level, "load"))
        except Exception as ex:   # noqa: BLE001
            errors.append((k, repr(ex)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ths), "a thread is stuck"
    assert not errors, errors[:5]


def test_concurrent_calls_really_overlap(gpu):
    """Two threads inside xpnghip_encode_tiles at the same time both finish, with the same bytes as a serial call, and the pair
    takes less than twice a single call (no lock is held across a call).  Timing assertion is loose: it only has to
    distinguish "serialised" from "side by side"."""
    import threading
    import time
    from xpng_amd import api
    from xpng_amd.synth import synth_raster
    rasters = [synth_raster("photo", 2048, 2048, True, seed=s) for s in (11, 12)]
    want = [api.encode_tiles(1, r) for r in rasters]
    api.encode_tiles(1, rasters[0]); api.encode_tiles(1, rasters[1])          # warm: contexts exist
    t0 = time.perf_counter()
    for r in rasters:
        api.encode_tiles(1, r)
    serial = time.perf_counter() - t0
    got = [None, None]

    def run(k):
        got[k] = api.encode_tiles(1, rasters[k])
    best = 1e9
    for _ in range(3):
        ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        best = min(best, time.perf_counter() - t0)
        assert got == want
    print(f"two encodes: serial {serial * 1e3:.1f} ms, concurrent {best * 1e3:.1f} ms")
    assert best < serial * 1.05


def test_first_call_on_a_fresh_context_may_be_anything(gpu, po):
    """Regression for the abort of gpurun_out/r2_t14.log: a context's lazily allocated buffers (the five
    symbol planes, the mode-2 workspace) are allocated by whichever launch sequence needs them first - a mode-2 encode, the
    transform-only entry, a mode-1 encode - and a launch never hands a null workspace pointer to a kernel
    (that is a GPU fault, i.e. abort(), not an error return)."""
    import torch
    import xpng_amd
    from xpng_amd.api import walk_tile_offsets
    from xpng_amd.synth import synth_raster
    raster = synth_raster("photo", 700, 500, False, seed=5)
    h, w, ch = raster.shape
    d_r = torch.from_numpy(raster).cuda()
    for first in ("m2_encode", "transform", "m2_decode", "m1_encode", "fetch"):
        ctx = xpng_amd.Context(w, h, ch)
        d_b = torch.zeros(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        if first == "m2_encode":
            n = ctx.encode_device(2, d_r.data_ptr(), d_b.data_ptr())
            assert d_b[:n].cpu().numpy().tobytes() == po.encode_tiles(2, raster)
        elif first == "transform":
            ctx.transform_device(d_r.data_ptr())
            t0_ = ctx.tiles()[0]
            pr, _ = po.choose_predictor(raster, t0_)
            assert np.array_equal(ctx.fetch("nl", 0), po.m1_planes(raster, t0_, pr)["nl"])
        elif first == "m2_decode":
            blobs = po.encode_tiles(2, raster)
            d_b[: len(blobs)] = torch.frombuffer(bytearray(blobs), dtype=torch.uint8).cuda()
            off, _ = walk_tile_offsets(blobs, ctx.n_tiles)
            d_o = torch.zeros(h * w * ch + 64, dtype=torch.uint8, device="cuda")
            ctx.decode_device(2, d_b.data_ptr(), len(blobs), off, d_o.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_o[: h * w * ch].cpu().numpy().reshape(h, w, ch), raster)
        elif first == "m1_encode":
            n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
            assert d_b[:n].cpu().numpy().tobytes() == po.encode_tiles(1, raster)
        else:
            with pytest.raises(xpng_amd.api.XpngError):
                ctx.fetch("nl", 0)                               # planes that do not exist yet: an error, not a null read
        ctx.close()


@pytest.mark.parametrize("force_wide", [False, True])
def test_context_stream_places_follow_their_lengths(gpu, po, monkeypatch, force_wide):
    """Stream-scratch layout (common.hpp): the nine context streams of a tile lie back to back, each in the room its length needs;
    the lengths come from k_m1_count (a histogram of the nl plane) BEFORE the routing kernel writes a byte.  Rasters that push
    the layout: nearly every pixel in one context (flat), all nine contexts busy (photo, noise), transparent runs (RGBA: coded
    pixels skip them), tiles of one row / one column, and a batch.  Every stream, k and every block byte-equal to the oracle."""
    import torch
    import xpng_amd
    from xpng_amd.synth import synth_raster
    if force_wide:
        monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    cases = [("photo", 700, 500, True), ("photo", 700, 500, False), ("noise", 300, 200, True), ("flat", 500, 460, False),
             ("photo", 1200, 3, False), ("photo", 5, 900, True), ("photo", 900, 460, "clear")]
    for kind, w, h, alpha in cases:
        clear = alpha == "clear"  # RGBA whose first tile codes NO pixel at all (every stream length 0) beside an ordinary one
        alpha = bool(alpha)
        raster = synth_raster(kind, w, h, alpha, seed=11)
        if clear:
            raster[:, :456, 3] = 0                     # (tile 0 of a 900 x 460 raster is 456 x 460)
        elif alpha:
            raster[h // 3: h // 3 + 7, :, 3] = 0          # runs of invisible pixels, also at the end of a row
            raster[-1, -max(1, w // 4):, 3] = 0           # the tile's LAST pixels are not coded: "last coded pixel" is further up
        hh, ww, ch = raster.shape
        ctx = xpng_amd.Context(ww, hh, ch)
        d_r = torch.from_numpy(raster).cuda()
        d_b = torch.zeros(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        n = ctx.encode_device(1, d_r.data_ptr(), d_b.data_ptr())
        assert d_b[:n].cpu().numpy().tobytes() == po.encode_tiles(1, raster), (kind, w, h, alpha)
        for ti, t in enumerate(ctx.tiles()):
            pr, _ = po.choose_predictor(raster, t)
            planes = po.m1_planes(raster, t, pr)
            st = po.m1_streams(raster, t, planes)
            for c in range(9):
                assert np.array_equal(ctx.fetch(10 + c, ti), st["ctx"][c]), (kind, w, h, alpha, ti, c)
            assert np.array_equal(ctx.fetch("k", ti).view(np.uint32), st["k"]), (kind, w, h, alpha, ti)
        ctx.close()
    # a batch of distinct rasters through one launch sequence
    rs = [synth_raster("photo", 900, 700, True, seed=20 + i) for i in range(5)]
    ctx = xpng_amd.Context(900, 700, 4, batch=5)
    d_rs = [torch.from_numpy(r).cuda() for r in rs]
    d_bs = [torch.zeros(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in rs]
    sizes = ctx.encode_device_batch(1, [d.data_ptr() for d in d_rs], [d.data_ptr() for d in d_bs])
    for r, d, n in zip(rs, d_bs, sizes):
        assert d[:n].cpu().numpy().tobytes() == po.encode_tiles(1, r)
    ctx.close()
@pytest.mark.gpu
@pytest.mark.parametrize("pb", [13, 15])
def test_context_stream_with_more_probability_bits_on_the_wide_path(gpu, po, monkeypatch, pb):
    """The reference writes its nl-context streams with 12 probability bits; the format allows 10..15 (libxpng.c:429-493 reads
    the width from the block header).  A file whose context blocks were re-coded with 13 / 15 bits by the oracle's encoder must
    decode to the same raster: on the wide path those streams do not fit the small table layout and go to k_rans2_decode_rest."""
    monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    from xpng_amd import api
    from xpng_amd.synth import synth_raster
    W, H = 1000, 300
    raster = synth_raster("photo", W, H, False, seed=11).copy()
    blobs = po.encode_tiles(1, raster)
    tiles = po.tile_table(W, H, 3)
    out, o, recoded = bytearray(), 0, 0
    for ti in range(len(tiles)):
        L = int.from_bytes(blobs[o:o + 3], "little")
        tile = bytearray(blobs[o:o + L])
        o += L
        if tile[3] == 0:                     # raw tile
            out += tile
            continue
        ksz = int.from_bytes(tile[4:8], "little")
        q, parts = 4 + ksz, [bytes(tile[:4 + ksz])]
        for c in range(9):
            b0 = int.from_bytes(tile[q:q + 4], "little")
            ty, sz = b0 >> 24, (4 if (b0 >> 24) == 0 else b0 & 0xFFFFFF)
            blk = bytes(tile[q:q + sz])
            q += sz
            if ty >= 3 and c % 2 == ti % 2:  # every other rANS-coded context block of the tile
                n = int.from_bytes(blk[4:7], "little")
                syms, _ = po.rans2_decode(blk, n)
                new = po.rans2_encode(np.bincount(syms, minlength=9).astype(np.uint32), 9, syms, pb)
                if new[3] >= 3 and new[11] == pb:
                    blk, recoded = new, recoded + 1
            parts.append(blk)
        assert q == L
        body = b"".join(parts)
        out += len(body).to_bytes(3, "little") + bytes([tile[3]]) + body[4:]
    assert recoded >= 4
    out = bytes(out)
    assert np.array_equal(po.decode_tiles(1, out, W, H, 3), raster)        # the crafted file is a valid one
    assert np.array_equal(api.decode_tiles(1, out, W, H, 3), raster)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(1003, 777, 4), (453, 130, 4), (70, 200, 4), (1003, 777, 3), (17, 300, 4)])
def test_band_reconstruction_row_staging_at_every_alignment(gpu, po, monkeypatch, geom):
    """The wide decode's band reconstruction moves rows through LDS in 64-byte-aligned chunks (recon_band_core): chunk phases depend
    on where a tile row starts in the raster and in the residual plane, and a row's first / last chunk leave word by word.  Odd
    widths (row pitch not a multiple of 64 bytes), tiles narrower than one chunk, and destination rasters at each 16-byte phase of a
    64-byte line: decode on the device, compare with the input raster, and check that nothing around the raster
    was written."""
    import torch
    monkeypatch.setenv("XPNG_WIDE_RANS", "1")
    from xpng_amd.synth import synth_raster
    W, H, ch = geom
    raster = synth_raster("photo", W, H, ch == 4, seed=W + H).copy()
    if ch == 4:
        raster[raster[..., 3] == 0] = 0
    blobs = po.encode_tiles(1, raster)
    ctx = gpu.Context(W, H, ch)
    d_blob = torch.from_numpy(np.frombuffer(blobs + b"\0" * 64, dtype=np.uint8).copy()).cuda()
    nbytes = W * H * ch
    for shift in (0, 16, 32, 48):   # (the C-ABI asks for 16-byte aligned device buffers; the odd row pitch supplies every 4-byte phase)
        buf = torch.full((nbytes + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        ctx.decode_device(1, d_blob.data_ptr(), len(blobs), None, buf.data_ptr() + 64 + shift)
        torch.cuda.synchronize()
        assert ctx.decode_status() == 0
        out = buf.cpu().numpy()
        assert np.array_equal(out[64 + shift:64 + shift + nbytes].reshape(H, W, ch), raster), (geom, shift)
        assert (out[:64 + shift] == 0xA5).all() and (out[64 + shift + nbytes:] == 0xA5).all(), (geom, shift)
    ctx.close()


@pytest.mark.parametrize("case", ["translucent_last_band", "hidden_late", "opaque", "rgb", "rgba_level2"])
def test_xpng_store_band_pipeline(gpu, po, tmp_path, case):
    """xpng_store on an ordinary large raster stages it in three tile-row bands, each on a stream of its own: upload ->
    normalize_RGBA flags -> hidden colours zeroed in place -> the band's tile encode (wrappers.hpp, staged image [r4]; the
    reference's serial pre-pass is libxpng.c:688-721, inside its timed region 727-760).  The only whole-image decision - the
    raster stays RGBA iff some pixel has alpha != 255 - is taken at the first band that holds such a pixel.  Odd widths put the
    bands at every 16-byte phase.  Every file equals the oracle's, and xpng_load returns the normalised raster."""
    from xpng_amd.synth import synth_raster
    w, h = 3001, 2999                     # 36 MB as RGBA, 27 MB as RGB: seven tile rows, three bands
    level = 1
    if case == "rgb":
        r = synth_raster("photo", w, h, False, seed=5)
        level = 2
    else:
        r = synth_raster("photo", w, h, True, seed=5)
        if case == "opaque":
            r[..., 3] = 255                                   # no translucent pixel anywhere: repacked to RGB after the last band
        elif case == "translucent_last_band":
            r[..., 3] = 255
            r[h - 3, w - 2, 3] = 254                          # ... except one, in the last band: stays RGBA, decided late
        elif case == "hidden_late":
            r[..., 3] = 255
            r[5, 7, 3] = 17                                   # band 0 settles "RGBA" at once; hidden colours exist only further down
            r[h // 2, 11] = (9, 8, 7, 0)
            r[h - 1, w - 1] = (1, 0, 0, 0)
        elif case == "rgba_level2":
            level = 2                                         # RGBA at level 2: the single-colour test runs first (whole raster), then level 1
    want = po.encode_image(level, r)
    out = tmp_path / "o.xpng"
    gpu.store(level, r, str(out))
    data = out.read_bytes()
    assert len(data) == len(want) and data == want, case
    back = gpu.load(str(out))
    norm = po.normalize_rgba(r) if r.shape[2] == 4 else r
    assert back.shape == norm.shape and np.array_equal(back, norm), case
    # the same raster again through a pooled staging object (buffers and streams reused), and a small one behind it (whole-image form)
    gpu.store(level, r, str(out))
    assert out.read_bytes() == want
    small = np.ascontiguousarray(r[:300, :400])
    gpu.store(level, small, str(out))
    assert out.read_bytes() == po.encode_image(level, small)
