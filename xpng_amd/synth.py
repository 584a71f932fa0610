"""Deterministic synthetic rasters for parity tests and bench.py.

No RNG library: every pixel is a pure integer hash of (x, y, channel, seed), so the same
bytes come out of numpy on the host and of torch int64 arithmetic on the GPU.  The md5s of
the files these generators produce are pinned in tests/test_synth.py against SURVEY.md §8(d).

kinds
  photo : smooth ramps + 3 bits of hash noise; RGBA alpha is chosen per 64x64 block
          (transparent-and-zero / gradient / opaque) so `normalize_RGBA` keeps the raster RGBA.
  noise : incompressible (forces raw tiles and the whole-file level-7 fallback).
  flat  : one colour (single-colour paths).
  gray  : R == G == B photo-like ramp (mode-2 grayscale tiles).
"""
from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF


def _hash_np(x, y, c, seed):
    h = (x.astype(np.uint64) * 0x9E3779B1 + y.astype(np.uint64) * 0x85EBCA77
         + np.uint64(c * 0xC2B2AE3D + seed)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * 0x2C1B3C6D) & _M32
    h ^= h >> np.uint64(12)
    h = (h * 0x297A2D39) & _M32
    h ^= h >> np.uint64(15)
    return h


def synth_raster(kind: str, w: int, h: int, alpha: bool, seed: int = 1,
                 x0: int = 0, y0: int = 0) -> np.ndarray:
    """Return an (h, w, 3+alpha) uint8 raster. (x0, y0) offsets let a rank build only its band."""
    ys, xs = np.meshgrid(np.arange(y0, y0 + h, dtype=np.uint64),
                         np.arange(x0, x0 + w, dtype=np.uint64), indexing="ij")
    ch = 3 + int(alpha)
    out = np.empty((h, w, ch), dtype=np.uint8)
    if kind == "photo":
        for c in range(3):
            v = ((xs * (c + 2) + ys * (5 - c)) >> np.uint64(3)) + (_hash_np(xs, ys, c, seed) & 7) + 64 * c
            out[..., c] = ((v.astype(np.int64) - 4) & 255).astype(np.uint8)
        if alpha:
            t = _hash_np(xs >> np.uint64(6), ys >> np.uint64(6), 7, seed) & 3
            a = np.where(t == 1, ((xs + ys) & 255) | 1, 255).astype(np.uint8)
            a = np.where(t == 0, 0, a).astype(np.uint8)
            out[..., 3] = a
            out[t == 0] = 0
    elif kind == "noise":
        for c in range(3):
            out[..., c] = (_hash_np(xs, ys, c, seed) >> np.uint64(24)).astype(np.uint8)
        if alpha:
            out[..., 3] = ((_hash_np(xs, ys, 3, seed) >> np.uint64(24)) | 1).astype(np.uint8)
    elif kind == "flat":
        out[...] = np.array([0x4D, 0x4D, 0x4D, 0x80][:ch], dtype=np.uint8)
    elif kind == "gray":
        v = (((xs * 3 + ys * 2) >> np.uint64(3)) + (_hash_np(xs, ys, 0, seed) & 3)) & 255
        for c in range(3):
            out[..., c] = v.astype(np.uint8)
        if alpha:
            out[..., 3] = 255
    else:
        raise ValueError(f"unknown synthetic kind {kind!r}")
    return out


def synth_raster_torch(kind: str, w: int, h: int, alpha: bool, seed: int = 1,
                       x0: int = 0, y0: int = 0, device="cuda", rows_per_chunk: int = 1024):
    """Same bytes as synth_raster, generated on `device` with int64 arithmetic (bench input is
    produced straight into HBM; a 16384^2 RGBA raster is 1 GiB)."""
    import torch

    ch = 3 + int(alpha)
    out = torch.empty((h, w, ch), dtype=torch.uint8, device=device)

    def hsh(x, y, c):
        v = (x * 0x9E3779B1 + y * 0x85EBCA77 + (c * 0xC2B2AE3D + seed)) & _M32
        v = v ^ (v >> 15)
        v = (v * 0x2C1B3C6D) & _M32
        v = v ^ (v >> 12)
        v = (v * 0x297A2D39) & _M32
        v = v ^ (v >> 15)
        return v

    xs = torch.arange(x0, x0 + w, dtype=torch.int64, device=device)[None, :]
    for r0 in range(0, h, rows_per_chunk):
        r1 = min(h, r0 + rows_per_chunk)
        ys = torch.arange(y0 + r0, y0 + r1, dtype=torch.int64, device=device)[:, None]
        o = out[r0:r1]
        if kind == "photo":
            for c in range(3):
                v = ((xs * (c + 2) + ys * (5 - c)) >> 3) + (hsh(xs, ys, c) & 7) - 4 + 64 * c
                o[..., c] = (v & 255).to(torch.uint8)
            if alpha:
                t = hsh(xs >> 6, ys >> 6, 7) & 3
                a = torch.where(t == 1, ((xs + ys) & 255) | 1, torch.full_like(t, 255))
                a = torch.where(t == 0, torch.zeros_like(a), a)
                o[..., 3] = a.to(torch.uint8)
                o[t == 0] = 0
        elif kind == "noise":
            for c in range(3):
                o[..., c] = (hsh(xs, ys, c) >> 24).to(torch.uint8)
            if alpha:
                o[..., 3] = ((hsh(xs, ys, 3) >> 24) | 1).to(torch.uint8)
        elif kind == "flat":
            o[...] = torch.tensor([0x4D, 0x4D, 0x4D, 0x80][:ch], dtype=torch.uint8, device=device)
        elif kind == "gray":
            v = (((xs * 3 + ys * 2) >> 3) + (hsh(xs, ys, 0) & 3)) & 255
            for c in range(3):
                o[..., c] = v.to(torch.uint8)
            if alpha:
                o[..., 3] = 255
        else:
            raise ValueError(f"unknown synthetic kind {kind!r}")
    return out


def seven_header(w: int, h: int, alpha: bool, level: int = 7) -> bytes:
    """8-byte `.7` / `.xpng` header (reference 7/libseven.c:11, libxpng.c:736)."""
    import struct
    return struct.pack("<II", (w - 1) | (level << 24), (h - 1) | (int(alpha) << 24))


def to_seven_bytes(raster: np.ndarray) -> bytes:
    h, w, ch = raster.shape
    return seven_header(w, h, ch == 4) + raster.tobytes()


def load_seven(path: str) -> np.ndarray:
    """.7 file -> (h, w, ch) uint8 raster (reference 7/libseven.c:18-36)."""
    import struct
    b = open(path, "rb").read()
    h0, h1 = struct.unpack("<II", b[:8])
    w, h, a = (h0 & 0xFFFFFF) + 1, (h1 & 0xFFFFFF) + 1, (h1 >> 24) & 1
    if h0 >> 24 != 7 or len(b) != 8 + w * h * (3 + a):
        raise ValueError(f"{path}: not a .7 file")
    return np.frombuffer(b, dtype=np.uint8, offset=8).reshape(h, w, 3 + a).copy()


def special_cases():
    """Named rasters that force the rarely taken branches (normalisation rewrite, opaque alpha
    drop, mixed gray / colour / single-colour tiles, raw gray tiles)."""
    r = synth_raster("photo", 300, 200, True); r[10:20, 10:20, 3] = 0; r[10:20, 10:20, 0] = 9
    yield "hidden_colour", r
    r = synth_raster("photo", 300, 200, True); r[..., 3] = 255
    yield "opaque_alpha", r
    r = synth_raster("photo", 1000, 900, False); r[:444, :500] = synth_raster("gray", 500, 444, False); r[444:, 500:] = 77
    yield "mixed_tiles", r
    r = synth_raster("gray", 900, 500, False); r[:, :450] = synth_raster("noise", 450, 500, False)[..., :1]
    yield "gray_noise", r
