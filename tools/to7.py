#!/usr/bin/env python3
"""PNG -> .7 converter (stand-in for the reference's 7/seven --to_7, which needs libpng headers).

RGBA if the PNG has alpha/tRNS else RGB, then the normalize_RGBA rule (reference 7/seven.c:4-37 ==
libxpng.c:688-721): zero RGB under alpha==0; drop a fully opaque alpha channel.
"""
import struct
import sys

import numpy as np
from PIL import Image


def png_to_raster(path: str) -> np.ndarray:
    im = Image.open(path)
    has_alpha = im.mode in ("RGBA", "LA", "PA") or "transparency" in im.info
    a = np.asarray(im.convert("RGBA" if has_alpha else "RGB")).copy()
    if a.shape[2] == 4:
        al = a[..., 3]
        if ((al == 0) & (a[..., :3].any(axis=2))).any():
            a[al == 0] = 0
        elif (al == 255).all():
            a = np.ascontiguousarray(a[..., :3])
    return a


def raster_to_seven(a: np.ndarray) -> bytes:
    h, w, ch = a.shape
    return struct.pack("<II", (w - 1) | (7 << 24), (h - 1) | ((ch - 3) << 24)) + a.tobytes()


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: to7.py in.png out.7")
    with open(sys.argv[2], "wb") as f:
        f.write(raster_to_seven(png_to_raster(sys.argv[1])))
