import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


from xpng_amd.synth import load_seven  # noqa: E402,F401


def golden_raster(name, ent):
    """Rebuild the input raster of a manifest entry (committed .7, synthetic generator or special)."""
    from xpng_amd.synth import special_cases, synth_raster
    src = ent["src"]
    if src.startswith("file:"):
        return load_seven(os.path.join(GOLD, src[5:]))
    if src.startswith("synth:"):
        return synth_raster(src[6:], ent["w"], ent["h"], ent["ch"] == 4)
    if src == "synth-special":
        for n, r in special_cases():
            if "special_" + n == name:
                return r
    return None


def small_entries(manifest, max_px=1_200_000):
    return [(n, e) for n, e in sorted(manifest.items())
            if e["w"] * e["h"] <= max_px and not e["src"].startswith("reference-corpus")]


def corpus_entries(manifest):
    """The reference's own test set (test.rb:28-38 over images/*.png, BASELINE config 5): 17 images, each with the
    reference-written level-1 and level-2 .xpng committed whole (RGBA images: one file, level 2 falls back to level 1)."""
    out = [(n, e) for n, e in sorted(manifest.items())
           if (n.startswith("imgfull_") or n in ("img_juicy", "img_pigz-logo")) and "file" in e.get("L1", {}) and "file" in e.get("L2", {})]
    assert len(out) == 17, len(out)
    return out


def corpus_raster(ent):
    """Input raster of a corpus entry: decode of the reference-written level-2 golden by the ORACLE (checker), pinned to the
    md5 of the .7 the reference was given."""
    import hashlib
    from oracle import pyoracle as po
    from xpng_amd.synth import to_seven_bytes
    data = open(os.path.join(GOLD, ent["L2"]["file"]), "rb").read()
    assert hashlib.md5(data).hexdigest() == ent["L2"]["md5"]
    r = po.decode_image(data)
    assert hashlib.md5(to_seven_bytes(r)).hexdigest() == ent["seven_md5"] == ent["L2"]["decoded_md5"]
    return r
