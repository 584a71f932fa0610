#!/usr/bin/env python3
"""Generate tests/golden/ from the GENUINE reference (oracle/_ref/xpng, compiled from /root/reference
by oracle/Makefile).  Container-only; the outputs (small .7 inputs, golden .xpng, manifest.json) are
committed, the reference itself never enters the repo.

  python oracle/make_golden.py          # corpus + synthetic edge set (+ 4096^2 md5s)
  python oracle/make_golden.py --big    # additionally the 16384^2 photo RGBA md5 (slow, ~1 GiB)
  python oracle/make_golden.py --only img_   # regenerate only the entries with that name prefix

manifest.json: name -> {w,h,ch, src: "file:<name>.7" | "synth:<kind>", seven_md5,
                        L1/L2/L7: {size, md5, file?}}
A golden .xpng is stored as a file when it is < 120 KB, otherwise only size+md5 are pinned -- except the 17 corpus images of the
reference's test.rb, whose level-1 and level-2 files are all committed whole (BASELINE config 5).
"""
import glob
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tools.to7 import png_to_raster, raster_to_seven  # noqa: E402
from xpng_amd.synth import special_cases, synth_raster  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
KEEP_XPNG_BELOW = 120_000

# synthetic edge set: (kind, w, h, alpha)
SYNTH = [(k, w, h, a)
         for (w, h) in [(1, 1), (2, 1), (1, 5), (3, 3), (4, 4), (5, 7), (17, 4), (64, 64), (444, 444), (445, 444),
                        (443, 445), (100, 2000), (2000, 100), (700, 500), (667, 667), (889, 445), (1000, 300)]
         for k in ("photo", "noise", "flat", "gray") for a in (False, True)]
BIG = [("photo", 4096, 4096, True), ("photo", 4096, 4096, False), ("noise", 4096, 4096, True), ("noise", 4096, 4096, False)]
# corpus images small enough to commit whole as .7; the rest are committed as crops
WHOLE = ["pigz-logo", "juicy"]
CROPS = {"2021": (100, 60, 520, 450), "olaf": (0, 300, 320, 240), "body_shop": (40, 200, 320, 240),
         "idiocracy": (300, 300, 320, 240), "Anomaly_in_the_Netherlands": (1500, 700, 320, 240),
         "unicorn": (500, 200, 320, 240), "pe4en_k": (100, 50, 320, 240), "rodina": (400, 0, 320, 240),
         "evil": (200, 100, 320, 240), "faster_horse": (300, 300, 320, 240)}


def md5(b):
    return hashlib.md5(b).hexdigest()


def main():
    assert po.have_ref(), "make -C oracle ref first"
    os.makedirs(GOLD, exist_ok=True)
    man = {}
    # --only PREFIX: regenerate just the entries whose name starts with PREFIX, keep the others from the committed manifest
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    old_man = json.load(open(os.path.join(GOLD, "manifest.json"))) if only else {}

    def add(name, raster, src, store_seven, keep_all=False):
        if only and not name.startswith(only):
            if name in old_man:
                man[name] = old_man[name]
            return
        seven = raster_to_seven(raster)
        h, w, ch = raster.shape
        ent = {"w": w, "h": h, "ch": ch, "src": src, "seven_md5": md5(seven)}
        if store_seven:
            with open(os.path.join(GOLD, name + ".7"), "wb") as f:
                f.write(seven)
        with tempfile.TemporaryDirectory() as td:
            for level in (1, 2, 7):
                if ch == 4 and (w < 4 or h < 4) and level != 7:
                    continue  # undefined behaviour in the reference
                out, _ = po.ref_encode(level, seven, td)
                back, _ = po.ref_decode(out, td)
                e = {"size": len(out), "md5": md5(out), "decoded_md5": md5(back)}
                if level != 7 and (keep_all or len(out) < KEEP_XPNG_BELOW):
                    fn = f"{name}.L{level}.xpng"
                    if level == 2 and ent.get("L1", {}).get("md5") == e["md5"] and "file" in ent["L1"]:
                        fn = ent["L1"]["file"]   # RGBA: level 2 falls back to level 1 (libxpng.c:755), same bytes: one file
                    else:
                        with open(os.path.join(GOLD, fn), "wb") as f:
                            f.write(out)
                    e["file"] = fn
                ent[f"L{level}"] = e
        man[name] = ent
        print(name, {k: v["size"] for k, v in ent.items() if k.startswith("L")}, flush=True)

    for p in sorted(glob.glob("/root/reference/images/*.png")):
        name = os.path.basename(p)[:-4]
        r = png_to_raster(p)
        if name in WHOLE:
            add("img_" + name, r, f"file:img_{name}.7", True, keep_all=True)
        else:  # pin the whole image by md5 only (input is not committed) ...
            # BASELINE config 5 (test.rb:28-38 image set): both golden .xpng are committed whole; the tests get the input
            # raster back by decoding the golden with the oracle and checking seven_md5
            add("imgfull_" + name, r, "reference-corpus (not committed)", False, keep_all=True)
        if name in CROPS:  # ... and a committed crop
            x, y, w, h = CROPS[name]
            c = np.ascontiguousarray(r[y:y + h, x:x + w])
            if c.shape[2] == 4 and (c[..., 3] == 255).all():
                c = np.ascontiguousarray(c[..., :3])
            add("crop_" + name, c, f"file:crop_{name}.7", True)
    for kind, w, h, a in SYNTH:
        add(f"synth_{kind}_{w}x{h}_{'rgba' if a else 'rgb'}", synth_raster(kind, w, h, a), f"synth:{kind}", False)
    for name, r in special_cases():
        add("special_" + name, r, "synth-special", False)
    for kind, w, h, a in BIG:
        add(f"synth_{kind}_{w}x{h}_{'rgba' if a else 'rgb'}", synth_raster(kind, w, h, a), f"synth:{kind}", False)
    if "--big" in sys.argv:
        W = H = 16384
        with tempfile.TemporaryDirectory() as td:
            src, dst = os.path.join(td, "in.7"), os.path.join(td, "out.xpng")
            hm = hashlib.md5()
            with open(src, "wb") as f:
                hdr = raster_to_seven(np.zeros((1, 1, 4), np.uint8))[:0]
                import struct
                hdr = struct.pack("<II", (W - 1) | (7 << 24), (H - 1) | (1 << 24))
                f.write(hdr); hm.update(hdr)
                for y0 in range(0, H, 512):
                    b = synth_raster("photo", W, 512, True, y0=y0).tobytes()
                    f.write(b); hm.update(b)
            import subprocess
            subprocess.check_call([po.REF_BIN, "-1", src, dst])
            out = open(dst, "rb").read()
            man["synth_photo_16384x16384_rgba"] = {"w": W, "h": H, "ch": 4, "src": "synth:photo", "seven_md5": hm.hexdigest(),
                                                   "L1": {"size": len(out), "md5": md5(out)}}
            print("16384^2", len(out), md5(out))
    elif os.path.exists(os.path.join(GOLD, "manifest.json")):
        old = json.load(open(os.path.join(GOLD, "manifest.json")))
        if "synth_photo_16384x16384_rgba" in old:
            man["synth_photo_16384x16384_rgba"] = old["synth_photo_16384x16384_rgba"]
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("entries:", len(man))


if __name__ == "__main__":
    main()
