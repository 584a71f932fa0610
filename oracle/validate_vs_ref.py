#!/usr/bin/env python3
"""Container-only check: the C restatement (libxpng_oracle.so) against the genuine compiled reference
(oracle/_ref/xpng) -- byte-identical .xpng for levels 1/2/7 and round trips, on the reference's own
image corpus and on synthetic edge rasters.  Needs /root/reference (for images) and oracle/_ref.
Usage: python oracle/validate_vs_ref.py [--quick]
"""
import glob
import hashlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from tools.to7 import png_to_raster, raster_to_seven  # noqa: E402
from xpng_amd.synth import synth_raster  # noqa: E402


def edge_rasters(quick):
    from xpng_amd.synth import special_cases
    rng_sizes = [(1, 1), (2, 1), (1, 5), (3, 3), (4, 4), (5, 7), (17, 4), (64, 64), (444, 444), (445, 444), (443, 445),
                 (100, 2000), (2000, 100), (700, 500), (667, 667), (889, 445), (1000, 300)]
    if not quick:
        rng_sizes += [(1334, 265), (1500, 1200), (300, 4000)]
    for (w, h) in rng_sizes:
        for kind in ("photo", "noise", "flat", "gray"):
            for alpha in (False, True):
                yield f"{kind}_{w}x{h}_{'rgba' if alpha else 'rgb'}", synth_raster(kind, w, h, alpha)
    yield from special_cases()


def main():
    quick = "--quick" in sys.argv
    assert po.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    bad = 0
    cases = []
    for p in sorted(glob.glob("/root/reference/images/*.png")):
        cases.append((os.path.basename(p)[:-4], png_to_raster(p)))
    cases += list(edge_rasters(quick))
    with tempfile.TemporaryDirectory() as td:
        for name, raster in cases:
            seven = raster_to_seven(raster)
            for level in (1, 2, 7):
                h, w, ch = raster.shape
                if ch == 4 and (w < 4 or h < 4) and level != 7:
                    continue  # reference undefined
                ref, _ = po.ref_encode(level, seven, td)
                mine = po.encode_image(level, raster, threads=0)
                ok = ref == mine
                back = po.decode_image(ref)
                nrm = po.normalize_rgba(raster)
                ok2 = back.shape == nrm.shape and np.array_equal(back, nrm)
                refback, _ = po.ref_decode(mine, td) if ok else (b"", "")
                ok3 = (not ok) or refback == raster_to_seven(nrm)
                if not (ok and ok2 and ok3):
                    bad += 1
                print(f"{'OK ' if ok and ok2 and ok3 else 'BAD'} {name:40s} L{level} {w}x{h}x{ch} ref={len(ref):9d} mine={len(mine):9d} "
                      f"md5={hashlib.md5(ref).hexdigest()[:16]} enc={ok} dec={ok2} refdec={ok3}", flush=True)
    print("FAILURES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
