#!/usr/bin/env python3
"""Orientation search (reference Mirroring_and_Rotating/test.rb): encode the 8 flips / rotations of an image and list the
sizes, smallest first.  The reference runs `tool` + `xpng -2` eight times; here the variants are made on the GPU
(torch.flip / rot90 of the device raster) and each geometry's variants go through ONE batched launch sequence of the tile
codec (level 2 for RGB, level 1 for RGBA, as xpng_store falls back: libxpng.c:755).

usage: orient_search.py image.7|image.png [level]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def variants(r):
    """(label, raster) for the reference's eight orientations; r90 = clockwise (tool.c op_r90)."""
    import torch
    out = []
    cur = r
    for rot in ("   0", "  90", " 180", " 270"):
        ms = ("    ", " + v", " + h") if rot in ("   0", "  90") else ("    ",)
        for m in ms:
            v = cur if m == "    " else (torch.flip(cur, [0]) if m == " + v" else torch.flip(cur, [1]))
            out.append((rot + m, v.contiguous()))
        cur = torch.rot90(cur, k=-1, dims=(0, 1)).contiguous()
    return out


def main():
    import torch
    import xpng_amd
    from xpng_amd.synth import load_seven
    path = sys.argv[1]
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if path.endswith(".png"):
        from PIL import Image
        im = Image.open(path)
        r = np.asarray(im.convert("RGBA" if "A" in im.getbands() or "transparency" in im.info else "RGB"))
    else:
        r = load_seven(path)
    d = torch.from_numpy(np.ascontiguousarray(r)).cuda()
    if d.shape[2] == 4:  # normalize_RGBA on the device, as xpng_store does
        out = torch.empty(d.numel() + 64, dtype=torch.uint8, device="cuda")
        pxsz, rewritten = xpng_amd.normalize_device(d.data_ptr(), d.shape[0] * d.shape[1], out.data_ptr())
        if rewritten:
            d = out[: d.shape[0] * d.shape[1] * pxsz].view(d.shape[0], d.shape[1], pxsz).clone()
    ch = d.shape[2]
    mode = 2 if (level == 2 and ch == 3) else 1
    vs = variants(d)
    sizes = {}
    by_geom = {}
    for label, v in vs:
        by_geom.setdefault((v.shape[1], v.shape[0]), []).append((label, v))
    for (w, h), lst in by_geom.items():
        ctx = xpng_amd.Context(w, h, ch, batch=len(lst))
        blobs = [torch.empty(ctx.blob_bound() + 64, dtype=torch.uint8, device="cuda") for _ in lst]
        lens = ctx.encode_device_batch(mode, [v.data_ptr() for _, v in lst], [b.data_ptr() for b in blobs])
        for (label, v), n in zip(lst, lens):
            raw = v.numel()
            sizes[label] = 8 + (n if n < raw else raw)   # "compressed >= raw -> level 7" rule of the driver (libxpng.c:771)
        ctx.close()
    width = max(len(str(s)) for s in sizes.values())
    print()
    for label, s in sorted(sizes.items(), key=lambda kv: kv[1]):
        print(f"\t{label}    =>    {s:{width}d}")
    print()
    return sizes


if __name__ == "__main__":
    main()
