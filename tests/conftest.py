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
