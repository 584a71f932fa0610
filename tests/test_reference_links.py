"""Container-only boundary proof: the reference's OWN command-line program, compiled from where it
lies under /root/reference with its own headers, links unchanged against this repo's libxpng.so and behaves as reference
xpng.c:3-24 says.  Skipped wherever /root/reference is absent (e.g. the GPU box); nothing of the reference is copied.
The level-7 / decode legs below never reach the tile codec, so they run without a GPU; the GPU leg is marked."""
import os
import subprocess

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "xpng_amd", "lib")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "xpng.c")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref_cli(tmp_path_factory):
    from xpng_amd import api
    api.host_lib()  # (fails loudly when libxpng.so has not been built)
    exe = str(tmp_path_factory.mktemp("refcli") / "xpng_ref_cli")
    # the command of INTEGRATION.md section A: the reference's xpng.c, this repo's libraries
    subprocess.check_call(["gcc", "-O2", os.path.join(REF, "xpng.c"), "-L" + LIB, "-lxpng", "-lxpng_hip", "-Wl,-rpath," + LIB, "-o", exe])
    return exe


def test_reference_cli_links_and_prints_its_usage(ref_cli):
    r = subprocess.run([ref_cli], capture_output=True, text=True)
    assert r.returncode == 1                                       # xpng.c:23
    assert "encode: ./xpng -[127] example.7    example.xpng" in r.stdout and "decode: ./xpng -d     example.xpng example.7" in r.stdout
    r = subprocess.run([ref_cli, "-3", "a.jpg", "b.xpng"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not Implemented." in r.stdout    # libxpng.c:1004-1014


def test_reference_cli_level7_and_decode_roundtrip(ref_cli, tmp_path):
    from xpng_amd.synth import synth_raster, to_seven_bytes
    for alpha in (False, True):
        src, mid, back = tmp_path / "a.7", tmp_path / "a.xpng", tmp_path / "b.7"
        seven = to_seven_bytes(synth_raster("photo", 77, 41, alpha))
        src.write_bytes(seven)
        assert subprocess.run([ref_cli, "-7", str(src), str(mid)]).returncode == 0
        assert mid.read_bytes() == seven                           # an .xpng at level 7 IS the .7 (7/libseven.c:11-15)
        assert subprocess.run([ref_cli, "-d", str(mid), str(back)]).returncode == 0
        assert back.read_bytes() == seven
    assert subprocess.run([ref_cli, "-d", str(tmp_path / "missing.xpng"), str(tmp_path / "x.7")]).returncode == 1


@pytest.mark.gpu
def test_reference_cli_drives_the_gpu_codec(ref_cli, tmp_path, manifest):
    """(only where a GPU and the reference tree are both present)"""
    from conftest import GOLD
    import hashlib
    src = os.path.join(GOLD, "img_pigz-logo.7")
    out, back = tmp_path / "o.xpng", tmp_path / "o.7"
    assert subprocess.run([ref_cli, "-1", src, str(out)]).returncode == 0
    assert hashlib.md5(out.read_bytes()).hexdigest() == manifest["img_pigz-logo"]["L1"]["md5"]
    assert subprocess.run([ref_cli, "-d", str(out), str(back)]).returncode == 0
    assert back.read_bytes() == open(src, "rb").read()
