"""CPU: the `seven` command (PNG <-> .7, reference 7/seven.c) binds the libpng16 runtime without headers."""
import ctypes.util
import os
import subprocess

import numpy as np
import pytest

from xpng_amd import api
from xpng_amd.synth import load_seven, synth_raster, to_seven_bytes

SEVEN = os.path.join(os.path.dirname(api.CLI), "seven")


@pytest.fixture(scope="module", autouse=True)
def built():
    api.build_native(("hip", "host"))
    if not (ctypes.util.find_library("png16") or os.path.exists("/usr/lib/x86_64-linux-gnu/libpng16.so.16")):
        pytest.skip("libpng16 runtime not installed")


def test_usage_and_exit_status():
    r = subprocess.run([SEVEN], capture_output=True, text=True)
    assert r.returncode == 1 and "--to_7" in r.stdout and "--to_png" in r.stdout


@pytest.mark.parametrize("kind,alpha", [("photo", True), ("photo", False), ("noise", True)])
def test_seven_png_seven_roundtrip_and_pixels(tmp_path, kind, alpha):
    from PIL import Image
    r = synth_raster(kind, 123, 77, alpha)
    a, png, b = tmp_path / "a.7", tmp_path / "a.png", tmp_path / "b.7"
    a.write_bytes(to_seven_bytes(r))
    assert subprocess.run([SEVEN, "--to_png", str(a), str(png)]).returncode == 0
    im = np.asarray(Image.open(png))
    assert im.shape == r.shape and np.array_equal(im, r)          # the PNG holds exactly the raster
    assert subprocess.run([SEVEN, "--to_7", str(png), str(b)]).returncode == 0
    assert b.read_bytes() == a.read_bytes()


def test_to_7_normalises_like_xpng_store(tmp_path):
    """Hidden colour under alpha 0 is zeroed; a fully opaque RGBA PNG becomes RGB (7/seven.c:4-37 = libxpng.c:688-721)."""
    from PIL import Image
    from oracle import pyoracle as po
    rng = np.random.default_rng(9)
    for case in ("hidden", "opaque"):
        r = synth_raster("noise", 61, 43, True)
        r[..., 3] = 255 if case == "opaque" else rng.integers(0, 3, (43, 61)) * 127
        png, out = tmp_path / f"{case}.png", tmp_path / f"{case}.7"
        Image.fromarray(r, "RGBA").save(png)
        assert subprocess.run([SEVEN, "--to_7", str(png), str(out)]).returncode == 0
        got = load_seven(str(out))
        want = np.ascontiguousarray(po.normalize_rgba(r))
        assert got.shape == want.shape and np.array_equal(got, want), case


@pytest.mark.skipif(not os.path.isdir("/root/reference/images"), reason="reference corpus not present (container-only test)")
def test_to_7_on_the_reference_corpus_gives_the_pinned_md5s():
    """SURVEY 8(c) / DESIGN 8: `seven --to_7` (host/seven_cli.c, the libpng16 runtime bound by dlopen) on the reference's own
    images/*.png (test.rb:28-30 feeds exactly these) produces the .7 files whose md5 the manifest pins as `seven_md5` - the
    inputs every corpus golden was generated from (reference 7/seven.c:39-79)."""
    import hashlib
    import json
    import tempfile
    from conftest import GOLD, corpus_entries
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    checked = 0
    with tempfile.TemporaryDirectory() as td:
        for name, ent in corpus_entries(man):
            stem = name.split("_", 1)[1]
            png = os.path.join("/root/reference/images", stem + ".png")
            assert os.path.exists(png), png
            out = os.path.join(td, "o.7")
            assert subprocess.run([SEVEN, "--to_7", png, out]).returncode == 0
            data = open(out, "rb").read()
            assert hashlib.md5(data).hexdigest() == ent["seven_md5"], stem
            assert len(data) == 8 + ent["w"] * ent["h"] * ent["ch"]
            checked += 1
    assert checked == 17
