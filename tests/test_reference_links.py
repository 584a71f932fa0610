"""Boundary proof: the reference's OWN command-line program, compiled from where it lies under
/root/reference with its own headers, links unchanged against this repo's libxpng.so and behaves as reference xpng.c:3-24 says.
Where /root/reference is present (the build container) the program is compiled afresh; elsewhere (the GPU box) the binary
oracle/_ref/xpng_refcli is used - built in the container by `make -C oracle ref` (git-ignored, travels with gpurun like
oracle/_ref/xpng), so the GPU leg runs with a GPU underneath.  Nothing of the reference's source is copied.
The level-7 / decode legs below never reach the tile codec, so they run without a GPU; the GPU leg is marked."""
import os
import subprocess

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "xpng_amd", "lib")

PREBUILT = os.path.join(ROOT, "oracle", "_ref", "xpng_refcli")


@pytest.fixture(scope="module")
def ref_cli(tmp_path_factory):
    from xpng_amd import api
    api.host_lib()  # (fails loudly when libxpng.so has not been built)
    if os.path.exists(os.path.join(REF, "xpng.c")):
        exe = str(tmp_path_factory.mktemp("refcli") / "xpng_ref_cli")
        # the command of INTEGRATION.md section A: the reference's xpng.c, this repo's libraries
        subprocess.check_call(["gcc", "-O2", os.path.join(REF, "xpng.c"), "-L" + LIB, "-lxpng", "-lxpng_hip", "-Wl,-rpath," + LIB, "-o", exe])
        return exe
    if os.access(PREBUILT, os.X_OK):
        return PREBUILT
    pytest.skip("neither the reference tree nor oracle/_ref/xpng_refcli is present")


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
    """The reference's own main() (xpng.c:3-24) -> xpng_store / xpng_load of this repo -> the HIP tile codec: bytes equal the
    reference-written goldens, for an RGBA and an RGB image and both rANS levels."""
    from conftest import GOLD
    import hashlib
    done = 0
    for name in ("img_pigz-logo", "img_juicy", "crop_pe4en_k", "crop_evil"):
        src = os.path.join(GOLD, name + ".7")
        if name not in manifest or not os.path.exists(src):
            continue
        for level in (1, 2):
            out, back = tmp_path / "o.xpng", tmp_path / "o.7"
            r = subprocess.run([ref_cli, f"-{level}", src, str(out)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert "encode," in r.stdout and "MPx/s" in r.stdout            # the line shape of libxpng.c:761-762
            assert hashlib.md5(out.read_bytes()).hexdigest() == manifest[name][f"L{level}"]["md5"], (name, level)
            assert subprocess.run([ref_cli, "-d", str(out), str(back)]).returncode == 0
            assert back.read_bytes() == open(src, "rb").read()
            done += 1
    assert done == 8
