"""ctypes binding of oracle/libxpng_oracle.so -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The shipped
package (xpng_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libxpng_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "xpng")


class XoTile(C.Structure):
    _fields_ = [("x", C.c_uint64), ("y", C.c_uint64), ("w", C.c_uint64), ("h", C.c_uint64)]


class XoStreams(C.Structure):
    _fields_ = [("ctx", C.POINTER(C.c_uint8) * 9), ("ctx_n", C.c_uint32 * 9),
                ("kwords", C.POINTER(C.c_uint32)), ("k_n", C.c_uint32),
                ("F", C.c_uint32 * 144), ("FA", C.c_uint32 * 256)]


_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "xpng_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u8p, u64 = C.POINTER(C.c_uint8), C.c_uint64
        L.xo_tile_table.restype = u64
        L.xo_tile_table.argtypes = [u64, u64, C.c_int, C.POINTER(XoTile), u64]
        L.xo_choose_predictor.restype = C.c_int
        L.xo_choose_predictor.argtypes = [C.c_void_p, u64, C.c_int, C.POINTER(XoTile), C.POINTER(C.c_uint32)]
        L.xo_m1_planes.restype = None
        L.xo_m1_planes.argtypes = [C.c_void_p, u64, C.c_int, C.POINTER(XoTile), C.c_int] + [C.c_void_p] * 5
        L.xo_m1_form_streams.restype = C.c_int
        L.xo_m1_form_streams.argtypes = [C.c_void_p, u64, C.c_int, C.POINTER(XoTile)] + [C.c_void_p] * 5 + [C.POINTER(XoStreams)]
        L.xo_m1_streams_free.argtypes = [C.POINTER(XoStreams)]
        L.xo_rans2_encode.restype = u64
        L.xo_rans2_encode.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, u64, C.c_void_p, C.c_int]
        L.xo_rans2_decode.restype = u64
        L.xo_rans2_decode.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(u64)]
        L.xo_tile_blob_bound.restype = u64
        L.xo_tile_blob_bound.argtypes = [C.POINTER(XoTile), C.c_int]
        L.xo_encode_tile.restype = u64
        L.xo_encode_tile.argtypes = [C.c_int, C.c_void_p, u64, C.c_int, C.POINTER(XoTile), C.c_void_p]
        L.xo_decode_tile.restype = C.c_int
        L.xo_decode_tile.argtypes = [C.c_int, C.c_void_p, C.c_void_p, u64, C.c_int, C.POINTER(XoTile)]
        L.xo_encode_image.restype = C.c_int
        L.xo_encode_image.argtypes = [C.c_int, C.c_void_p, u64, u64, C.c_int, C.POINTER(u8p), C.POINTER(u64), C.c_int]
        L.xo_decode_image.restype = C.c_int
        L.xo_decode_image.argtypes = [C.c_void_p, u64, C.POINTER(u8p), C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int), C.c_int]
        L.xo_encode_tiles.restype = C.c_int
        L.xo_encode_tiles.argtypes = [C.c_int, C.c_void_p, u64, u64, C.c_int, C.POINTER(u8p), C.POINTER(u64), C.c_int]
        L.xo_decode_tiles.restype = C.c_int
        L.xo_decode_tiles.argtypes = [C.c_int, C.c_void_p, u64, u64, u64, C.c_int, C.c_void_p, C.c_int]
        L.xo_normalize_rgba.restype = C.c_int
        L.xo_normalize_rgba.argtypes = [C.c_void_p, u64, u64, C.POINTER(u8p), C.POINTER(C.c_int)]
        L.xo_last_encode_ns.restype = u64
        L.xo_last_decode_ns.restype = u64
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def tile_table(w: int, h: int, pxsz: int):
    L = lib()
    n = L.xo_tile_table(w, h, pxsz, None, 0)
    arr = (XoTile * n)()
    L.xo_tile_table(w, h, pxsz, arr, n)
    return [(t.x, t.y, t.w, t.h) for t in arr]


def _tile(t):
    return XoTile(*t)


def choose_predictor(raster: np.ndarray, tile):
    h, w, ch = raster.shape
    sums = (C.c_uint32 * 4)()
    pr = lib().xo_choose_predictor(_ptr(raster), w, ch, C.byref(_tile(tile)), sums)
    return pr, list(sums)


def m1_planes(raster: np.ndarray, tile, pr: int):
    """-> dict of uint8 planes (nl, r, g, b, a) each of length tw*th."""
    h, w, ch = raster.shape
    n = tile[2] * tile[3]
    out = {k: np.zeros(n, dtype=np.uint8) for k in ("nl", "r", "g", "b", "a")}
    lib().xo_m1_planes(_ptr(raster), w, ch, C.byref(_tile(tile)), pr, _ptr(out["nl"]), _ptr(out["r"]),
                       _ptr(out["g"]), _ptr(out["b"]), _ptr(out["a"]) if ch == 4 else None)
    return out


def m1_streams(raster: np.ndarray, tile, planes):
    h, w, ch = raster.shape
    s = XoStreams()
    rc = lib().xo_m1_form_streams(_ptr(raster), w, ch, C.byref(_tile(tile)), _ptr(planes["nl"]), _ptr(planes["r"]),
                                  _ptr(planes["g"]), _ptr(planes["b"]), _ptr(planes["a"]) if ch == 4 else None, C.byref(s))
    assert rc == 0
    out = {
        "ctx": [np.ctypeslib.as_array(s.ctx[c], shape=(s.ctx_n[c],)).copy() if s.ctx_n[c] else np.zeros(0, np.uint8) for c in range(9)],
        "k": np.ctypeslib.as_array(s.kwords, shape=(s.k_n,)).copy(),
        "F": np.array(list(s.F), dtype=np.uint32).reshape(9, 16),
        "FA": np.array(list(s.FA), dtype=np.uint32),
    }
    lib().xo_m1_streams_free(C.byref(s))
    return out


def rans2_encode(F: np.ndarray, nominal_n: int, syms: np.ndarray, pb: int) -> bytes:
    F = np.ascontiguousarray(F, dtype=np.uint32).copy()
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    out = np.zeros(16 + 4 * len(syms) + 4 * 300, dtype=np.uint8)
    sz = lib().xo_rans2_encode(_ptr(F), nominal_n, _ptr(syms), len(syms), _ptr(out), pb)
    return out[:sz].tobytes()


def rans2_decode(block: bytes, max_n: int):
    buf = np.frombuffer(block + b"\0" * 16, dtype=np.uint8).copy()
    out = np.zeros(max_n + 8, dtype=np.uint8)
    n = C.c_uint64()
    csz = lib().xo_rans2_decode(_ptr(buf), _ptr(out), C.byref(n))
    return out[: n.value].copy(), csz


def encode_tile(mode: int, raster: np.ndarray, tile) -> bytes:
    h, w, ch = raster.shape
    t = _tile(tile)
    cap = lib().xo_tile_blob_bound(C.byref(t), ch)
    out = np.zeros(cap, dtype=np.uint8)
    sz = lib().xo_encode_tile(mode, _ptr(raster), w, ch, C.byref(t), _ptr(out))
    return out[:sz].tobytes()


def encode_tiles(mode: int, raster: np.ndarray, threads: int = 0) -> bytes:
    h, w, ch = raster.shape
    p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
    rc = lib().xo_encode_tiles(mode, _ptr(raster), w, h, ch, C.byref(p), C.byref(n), threads)
    if rc:
        raise RuntimeError("xo_encode_tiles failed")
    b = C.string_at(p, n.value)
    _libc.free(p)
    return b


def decode_tiles(mode: int, blobs: bytes, w: int, h: int, pxsz: int, threads: int = 0) -> np.ndarray:
    raster = np.zeros((h, w, pxsz), dtype=np.uint8)
    buf = np.frombuffer(blobs + b"\0" * 16, dtype=np.uint8)
    rc = lib().xo_decode_tiles(mode, _ptr(buf), len(blobs), w, h, pxsz, _ptr(raster), threads)
    if rc:
        raise RuntimeError("xo_decode_tiles failed")
    return raster


def encode_image(level: int, raster: np.ndarray, threads: int = 0) -> bytes:
    """raster (h, w, 3|4) uint8 -> .xpng file bytes (normalisation and fallbacks included)."""
    raster = np.ascontiguousarray(raster)
    h, w, ch = raster.shape
    p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
    rc = lib().xo_encode_image(level, _ptr(raster), w, h, ch - 3, C.byref(p), C.byref(n), threads)
    if rc:
        raise RuntimeError("xo_encode_image failed")
    b = C.string_at(p, n.value)
    _libc.free(p)
    return b


def decode_image(data: bytes, threads: int = 0) -> np.ndarray:
    buf = np.frombuffer(data + b"\0" * 16, dtype=np.uint8)
    p = C.POINTER(C.c_uint8)()
    w, h, a = C.c_uint64(), C.c_uint64(), C.c_int()
    rc = lib().xo_decode_image(_ptr(buf), len(data), C.byref(p), C.byref(w), C.byref(h), C.byref(a), threads)
    if rc:
        raise RuntimeError("xo_decode_image failed")
    ch = 3 + a.value
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, ch)).copy()
    _libc.free(p)
    return out


def normalize_rgba(raster: np.ndarray) -> np.ndarray:
    h, w, ch = raster.shape
    if ch == 3:
        return raster
    p, a = C.POINTER(C.c_uint8)(), C.c_int()
    rc = lib().xo_normalize_rgba(_ptr(np.ascontiguousarray(raster)), w, h, C.byref(p), C.byref(a))
    assert rc == 0
    if not p:
        return raster
    out = np.ctypeslib.as_array(p, shape=(h, w, 3 + a.value)).copy()
    _libc.free(p)
    return out


def last_encode_ns() -> int:
    return lib().xo_last_encode_ns()


def last_decode_ns() -> int:
    return lib().xo_last_decode_ns()


# ---------------------------------------------------------------- genuine reference (oracle/_ref)

def have_ref() -> bool:
    return os.access(REF_BIN, os.X_OK)


def ref_encode(level: int, seven_bytes: bytes, tmpdir: str, binary: str = REF_BIN):
    """Run the compiled reference CLI: .7 bytes -> (.xpng bytes, stdout)."""
    src, dst = os.path.join(tmpdir, "in.7"), os.path.join(tmpdir, "out.xpng")
    with open(src, "wb") as f:
        f.write(seven_bytes)
    r = subprocess.run([binary, f"-{level}", src, dst], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"reference encode failed rc={r.returncode}")
    with open(dst, "rb") as f:
        return f.read(), r.stdout


def ref_decode(xpng_bytes: bytes, tmpdir: str, binary: str = REF_BIN):
    src, dst = os.path.join(tmpdir, "in.xpng"), os.path.join(tmpdir, "out.7")
    with open(src, "wb") as f:
        f.write(xpng_bytes)
    r = subprocess.run([binary, "-d", src, dst], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"reference decode failed rc={r.returncode}")
    with open(dst, "rb") as f:
        return f.read(), r.stdout
