"""CPU: the C-ABI libraries build, load and export exactly what include/*.h declares (no compute calls)."""
import os
import re
import subprocess

import pytest

from xpng_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    api.build_native(("hip", "probes", "host"))


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


def test_release_library_has_no_timing_study_switches():
    """a product .so whose output can be falsified through the environment is not shippable.  The release
    library may read only the documented same-bytes form selectors; everything else lives in libxpng_hip_probes.so."""
    allowed = {"XPNG_DEVICE", "XPNG_GPUS", "XPNG_WIDE_RANS", "XPNG_NARROW_RANS", "XPNG_NO_SPLIT"}
    def names(so):
        out = subprocess.check_output(["strings", so], text=True)
        return {ln.strip() for ln in out.splitlines() if re.fullmatch(r"XPNG_[A-Z0-9_]+", ln.strip())}
    rel, prb = names(api.HIP_SO), names(api.PROBES_SO)
    assert rel <= allowed, sorted(rel - allowed)
    assert {"XPNG_SKIP", "XPNG_DBG_NOSTORE", "XPNG_FAKE_DEVICES", "XPNG_STAMPS", "XPNG_PAD_CHAIN"} <= prb
    ex = exported(api.PROBES_SO)
    assert not [n for n in api.HIP_SYMBOLS if n not in ex]
    assert api.hip_lib().xpnghip_probes_built() == 0 and api.probes_lib().xpnghip_probes_built() == 1
    assert api.hip_lib().xpnghip_debug_probe(None, 0) != 0      # the wave probe does not exist in the release library


def test_device_split_matches_the_rank_split():
    """The C multi-device wrappers (shard_ranges, csrc/wrappers.hpp) and the torch.distributed path (shard.py
    weighted_tile_ranges) cut an image into the same contiguous pixel-weighted tile ranges; no range is empty while tiles last
    (an empty shard made a wild host read)."""
    from xpng_amd.shard import tile_table, weighted_tile_ranges
    for (w, h) in [(4096, 4096), (16384, 16384), (1500, 1200), (445, 444), (300, 4000), (3799, 1927), (889, 445), (100, 100)]:
        tiles = tile_table(w, h)
        for D in (1, 2, 3, 4, 5, 7, 8, 16, 80, 81, 82, 200):
            got = api.shard_ranges(w, h, D)
            want = [r for r in weighted_tile_ranges(tiles, D) if r[0] < r[1] or len(tiles) == 0]
            assert got == want, (w, h, D)
            assert len(got) == min(D, len(tiles)) and got[0][0] == 0 and got[-1][1] == len(tiles)
            assert all(a < b for a, b in got) and all(x[1] == y[0] for x, y in zip(got, got[1:]))


def test_host_library_exports_reference_api():
    names = [n for n in declared("xpng.h", "xpng_") if n != "xpng_t"] + ["store_7", "load_7"]
    assert sorted(names) == sorted(api.HOST_SYMBOLS)
    ex = exported(api.HOST_SO)
    assert not [n for n in names if n not in ex]


def test_libraries_load_and_report_no_device_without_gpu():
    L = api.hip_lib()
    assert L.xpnghip_abi_version() == 2
    assert L.xpnghip_device_count() >= 0
    api.host_lib()


def test_gfx950_code_object_is_embedded():
    """The fat binary inside both flavours of the HIP library holds exactly one device code object, and it is gfx950's: the
    offload bundle's entry id (`hipv4-amdgcn-amd-amdhsa--gfx950`) is in the .hip_fatbin section, and no other gfx target is."""
    import re
    for so in (api.HIP_SO, api.PROBES_SO):
        blob = open(so, "rb").read()
        ids = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
        assert ids == {b"gfx950"}, (so, ids)


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
