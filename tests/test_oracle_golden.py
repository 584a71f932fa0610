"""CPU: the C restatement (oracle/) against the reference-generated golden vectors in tests/golden/.

Pins the oracle: every manifest entry whose input can be rebuilt here (committed .7, synthetic
generator) must encode to the byte-identical .xpng the compiled reference produced (size + md5, and
the stored file where present) and decode back to the reference's decoded .7.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, corpus_entries, corpus_raster, golden_raster, small_entries
from oracle import pyoracle as po
from xpng_amd.synth import to_seven_bytes


def md5(b):
    return hashlib.md5(b).hexdigest()


def test_manifest_has_corpus_and_edges(manifest):
    assert len(manifest) >= 150
    assert manifest["img_pigz-logo"]["L1"]["size"] == 43796
    assert manifest["img_pigz-logo"]["L1"]["md5"].startswith("f034ca0bcf8ccf68")
    assert manifest["synth_photo_4096x4096_rgba"]["L1"]["md5"].startswith("b697efb7f6d3")


def test_oracle_matches_reference_on_all_small_goldens(manifest):
    checked = 0
    for name, ent in small_entries(manifest):
        raster = golden_raster(name, ent)
        assert raster is not None, name
        assert md5(to_seven_bytes(raster)) == ent["seven_md5"], name
        for level in (1, 2, 7):
            g = ent.get(f"L{level}")
            if g is None:
                continue
            out = po.encode_image(level, raster)
            assert len(out) == g["size"] and md5(out) == g["md5"], (name, level)
            if "file" in g:
                assert out == open(os.path.join(GOLD, g["file"]), "rb").read()
            back = po.decode_image(out)
            assert md5(to_seven_bytes(back)) == g["decoded_md5"], (name, level)
            checked += 1
    assert checked >= 300


def test_thread_count_does_not_change_bytes(manifest):
    ent = manifest["crop_2021"]
    r = golden_raster("crop_2021", ent)
    assert po.encode_image(1, r, threads=1) == po.encode_image(1, r, threads=3)
    assert po.encode_image(2, r, threads=1) == po.encode_image(2, r, threads=4)


@pytest.mark.parametrize("name", ["synth_photo_4096x4096_rgba", "synth_photo_4096x4096_rgb"])
def test_oracle_matches_reference_4096(manifest, name):
    ent = manifest[name]
    raster = golden_raster(name, ent)
    for level in (1, 2):
        out = po.encode_image(level, raster)
        assert md5(out) == ent[f"L{level}"]["md5"], (name, level)
    back = po.decode_image(out)
    assert np.array_equal(back, raster)


def test_appendix_b_known_answer():
    """444x444 RGB of 4d4d4d at level 1 -> the 60 explained bytes."""
    r = np.full((444, 444, 3), 0x4D, dtype=np.uint8)
    out = po.encode_image(1, r)
    exp = bytes.fromhex("bb010001bb010000" "34000010" "08000000" "004d4d4d" "08000001" "0f020300" + "04000000" * 8)
    assert out == exp
    assert po.encode_image(2, r) == bytes.fromhex("bb010002bb0100024d4d4d")


def test_stage_functions_compose_to_tile_blob(manifest):
    """planes -> streams -> rANS blocks assembled by hand == xo_encode_tile (stage APIs are what the
    GPU kernels are checked against, so they must themselves be consistent with the pinned file path)."""
    import struct
    for name in ("img_pigz-logo", "crop_olaf"):
        raster = golden_raster(name, manifest[name])
        h, w, ch = raster.shape
        tiles = po.tile_table(w, h, ch)
        assert len(tiles) == 1
        t = tiles[0]
        pr, sums = po.choose_predictor(raster, t)
        planes = po.m1_planes(raster, t, pr)
        st = po.m1_streams(raster, t, planes)
        body = struct.pack("<I", 4 + 4 * len(st["k"])) + st["k"].tobytes()
        for c in range(9):
            body += po.rans2_encode(st["F"][c], 9, st["ctx"][c], 12)
        if ch == 4:
            body += po.rans2_encode(st["FA"], 256, planes["a"][1:], 15)
        blob = struct.pack("<I", (1 << 28) + (pr << 24) + len(body) + 4) + body
        assert blob == po.encode_tile(1, raster, t)
        assert blob == po.encode_image(1, raster)[8:]


def test_rans2_roundtrip_edge_blocks():
    rng = np.random.default_rng(7)
    for n, hi, pb in [(0, 9, 12), (1, 9, 12), (2, 9, 12), (3, 2, 12), (1000, 9, 12), (1001, 5, 12), (5000, 256, 15), (4097, 3, 15), (300, 256, 15)]:
        syms = (rng.integers(0, hi, n) if n else np.zeros(0)).astype(np.uint8)
        if n > 10:
            syms[rng.integers(0, n, n // 2)] = 0  # skew
        F = np.bincount(syms, minlength=256).astype(np.uint32)
        blk = po.rans2_encode(F, 256 if pb == 15 else 9, syms, pb)
        back, csz = po.rans2_decode(blk, n)
        assert csz == len(blk) and np.array_equal(back, syms), (n, hi, pb)


def test_oracle_on_the_whole_reference_corpus(manifest):
    """BASELINE config 5 image set (test.rb:28-38), whole images: the oracle decodes every reference-written golden to the
    reference's .7 and re-encodes it to the reference's bytes, levels 1 and 2."""
    for name, ent in corpus_entries(manifest):
        raster = corpus_raster(ent)
        assert raster.shape == (ent["h"], ent["w"], ent["ch"]), name
        for level in (1, 2):
            g = ent[f"L{level}"]
            gold = open(os.path.join(GOLD, g["file"]), "rb").read()
            assert len(gold) == g["size"] and md5(gold) == g["md5"], (name, level)
            assert md5(to_seven_bytes(po.decode_image(gold))) == g["decoded_md5"], (name, level)
            assert po.encode_image(level, raster) == gold, (name, level)
