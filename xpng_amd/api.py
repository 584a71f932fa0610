"""ctypes binding of libxpng_hip.so (include/xpng_hip.h) and libxpng.so (include/xpng.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "xpng_amd", "lib")
HIP_SO = os.path.join(LIBDIR, "libxpng_hip.so")
PROBES_SO = os.path.join(LIBDIR, "libxpng_hip_probes.so")  # -DXPNG_PROBES build: timing-study switches, wave probe, fake devices
HOST_SO = os.path.join(LIBDIR, "libxpng.so")
CLI = os.path.join(ROOT, "xpng_amd", "bin", "xpng")

# every symbol include/xpng_hip.h declares (tests/test_abi.py checks the .so exports all of them)
HIP_SYMBOLS = [
    "xpnghip_abi_version", "xpnghip_device_count", "xpnghip_last_error", "xpnghip_encode_tiles",
    "xpnghip_decode_tiles", "xpnghip_ctx_create", "xpnghip_ctx_destroy", "xpnghip_ctx_tile_count",
    "xpnghip_ctx_tile", "xpnghip_ctx_blob_bound", "xpnghip_ctx_workspace_bytes", "xpnghip_encode_device",
    "xpnghip_ctx_last_blobs_len", "xpnghip_decode_device", "xpnghip_m1_transform_device", "xpnghip_debug_fetch",
    "xpnghip_debug_probe", "xpnghip_debug_probe_count",
    "xpnghip_ctx_create_batch", "xpnghip_ctx_batch", "xpnghip_encode_device_batch", "xpnghip_ctx_last_blobs_len_at",
    "xpnghip_decode_device_batch", "xpnghip_m1_transform_device_batch", "xpnghip_ctx_create_range", "xpnghip_ctx_decode_status",
    "xpnghip_image_begin", "xpnghip_image_single_colour", "xpnghip_image_encode", "xpnghip_image_fetch", "xpnghip_image_end",
    "xpnghip_normalize_device", "xpnghip_encode_tiles_T", "xpnghip_decode_tiles_T", "xpnghip_image_encode_T", "xpnghip_devices_for",
    "xpnghip_shard_ranges", "xpnghip_shutdown", "xpnghip_probes_built",
]
HOST_SYMBOLS = ["xpng_store", "xpng_load", "xpng_from_jpg", "xpng_store_T", "xpng_load_T", "xpng_from_jpg_T",
                "store_7", "load_7"]


class XpngError(RuntimeError):
    pass


def native_paths():
    return {"hip": HIP_SO, "host": HOST_SO, "cli": CLI}


def build_native(targets=("hip", "host")) -> None:
    """Compile the native libraries in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", ROOT, *targets])


class XpngT(C.Structure):  # include/xpng.h xpng_t
    _fields_ = [("p", C.POINTER(C.c_uint8)), ("w", C.c_uint64), ("h", C.c_uint64), ("s", C.c_uint64), ("A", C.c_bool)]


_hip = None
_host = None
_probes = None


def _bind_hip(path):
    L = C.CDLL(path)
    if True:
        u64, vp = C.c_uint64, C.c_void_p
        L.xpnghip_abi_version.restype = C.c_int
        L.xpnghip_device_count.restype = C.c_int
        L.xpnghip_last_error.restype = C.c_char_p
        L.xpnghip_encode_tiles.restype = C.c_int
        L.xpnghip_encode_tiles.argtypes = [C.c_int, vp, u64, u64, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(u64)]
        L.xpnghip_decode_tiles.restype = C.c_int
        L.xpnghip_decode_tiles.argtypes = [C.c_int, vp, u64, u64, u64, C.c_int, vp]
        L.xpnghip_encode_tiles_T.restype = C.c_int
        L.xpnghip_encode_tiles_T.argtypes = [u64, C.c_int, vp, u64, u64, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(u64)]
        L.xpnghip_decode_tiles_T.restype = C.c_int
        L.xpnghip_decode_tiles_T.argtypes = [u64, C.c_int, vp, u64, u64, u64, C.c_int, vp]
        L.xpnghip_devices_for.restype = C.c_int
        L.xpnghip_devices_for.argtypes = [u64, u64, u64]
        L.xpnghip_ctx_create.restype = C.c_int
        L.xpnghip_ctx_create.argtypes = [C.POINTER(vp), C.c_int, u64, u64, C.c_int]
        L.xpnghip_ctx_create_batch.restype = C.c_int
        L.xpnghip_ctx_create_batch.argtypes = [C.POINTER(vp), C.c_int, u64, u64, C.c_int, C.c_uint32]
        L.xpnghip_ctx_create_range.restype = C.c_int
        L.xpnghip_ctx_create_range.argtypes = [C.POINTER(vp), C.c_int, u64, u64, C.c_int, C.c_uint32, u64, u64]
        L.xpnghip_ctx_batch.restype = C.c_uint32
        L.xpnghip_ctx_batch.argtypes = [vp]
        L.xpnghip_encode_device_batch.restype = C.c_int
        L.xpnghip_encode_device_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.c_uint32, u64, u64, C.POINTER(vp), C.POINTER(u64), vp]
        L.xpnghip_ctx_last_blobs_len_at.restype = u64
        L.xpnghip_ctx_last_blobs_len_at.argtypes = [vp, C.c_uint32]
        L.xpnghip_decode_device_batch.restype = C.c_int
        L.xpnghip_decode_device_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(u64), C.c_uint32, C.POINTER(u64), u64, u64, C.POINTER(vp), vp]
        L.xpnghip_ctx_decode_status.restype = C.c_int
        L.xpnghip_ctx_decode_status.argtypes = [vp, vp]
        L.xpnghip_ctx_destroy.restype = None
        L.xpnghip_ctx_destroy.argtypes = [vp]
        L.xpnghip_ctx_tile_count.restype = u64
        L.xpnghip_ctx_tile_count.argtypes = [vp]
        L.xpnghip_ctx_tile.restype = C.c_int
        L.xpnghip_ctx_tile.argtypes = [vp, u64, C.POINTER(u64)]
        L.xpnghip_ctx_blob_bound.restype = u64
        L.xpnghip_ctx_blob_bound.argtypes = [vp, u64, u64]
        L.xpnghip_ctx_workspace_bytes.restype = u64
        L.xpnghip_ctx_workspace_bytes.argtypes = [vp]
        L.xpnghip_encode_device.restype = C.c_int
        L.xpnghip_encode_device.argtypes = [vp, C.c_int, vp, u64, u64, vp, C.POINTER(u64), vp]
        L.xpnghip_ctx_last_blobs_len.restype = u64
        L.xpnghip_ctx_last_blobs_len.argtypes = [vp]
        L.xpnghip_decode_device.restype = C.c_int
        L.xpnghip_decode_device.argtypes = [vp, C.c_int, vp, u64, C.POINTER(u64), u64, u64, vp, vp]
        L.xpnghip_m1_transform_device.restype = C.c_int
        L.xpnghip_m1_transform_device.argtypes = [vp, vp, u64, u64, vp]
        L.xpnghip_m1_transform_device_batch.restype = C.c_int
        L.xpnghip_m1_transform_device_batch.argtypes = [vp, C.POINTER(vp), C.c_uint32, u64, u64, vp]
        L.xpnghip_debug_fetch.restype = C.c_int64
        L.xpnghip_debug_probe.argtypes = [vp, C.c_uint32]
        L.xpnghip_debug_probe_count.restype = C.c_int64
        L.xpnghip_debug_fetch.argtypes = [vp, C.c_int, u64, vp, u64]
        L.xpnghip_normalize_device.restype = C.c_int
        L.xpnghip_normalize_device.argtypes = [vp, u64, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]
        L.xpnghip_shard_ranges.restype = C.c_int
        L.xpnghip_shard_ranges.argtypes = [u64, u64, C.c_int, C.POINTER(u64), C.c_int]
        L.xpnghip_shutdown.restype = None
        L.xpnghip_probes_built.restype = C.c_int
        L.xpnghip_image_begin.restype = C.c_int
        L.xpnghip_image_begin.argtypes = [C.POINTER(vp), vp, u64, u64, C.c_int, C.POINTER(C.c_int)]
        L.xpnghip_image_single_colour.restype = C.c_int
        L.xpnghip_image_single_colour.argtypes = [vp, C.POINTER(C.c_int)]
        L.xpnghip_image_encode_T.restype = C.c_int
        L.xpnghip_image_encode_T.argtypes = [vp, u64, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(u64)]
        L.xpnghip_image_fetch.restype = C.c_int
        L.xpnghip_image_fetch.argtypes = [vp, vp]
        L.xpnghip_image_end.restype = None
        L.xpnghip_image_end.argtypes = [vp]
    return L


def hip_lib():
    """The release library.  XPNG_USE_PROBES_LIB=1 (tools/ only) makes every binding in this module use the probe build."""
    global _hip
    if _hip is None:
        path = PROBES_SO if os.environ.get("XPNG_USE_PROBES_LIB") else HIP_SO
        if not os.path.exists(path):
            raise XpngError(f"{path} is missing: run `make hip` (there is no CPU fallback)")
        _hip = _bind_hip(path)
    return _hip


def probes_lib():
    """libxpng_hip_probes.so (-DXPNG_PROBES) as a second, independent handle: for the tests / tools that need a switch the
    release library does not have (XPNG_FAKE_DEVICES, XPNG_SKIP, pads, the wave probe)."""
    global _probes
    if _probes is None:
        if not os.path.exists(PROBES_SO):
            raise XpngError(f"{PROBES_SO} is missing: run `make probes`")
        _probes = _bind_hip(PROBES_SO)
    return _probes


def shard_ranges(w: int, h: int, D: int):
    """Tile ranges a host-buffer call on D devices uses (xpnghip_shard_ranges; host-only, needs no GPU)."""
    arr = (C.c_uint64 * (2 * max(D, 1)))()
    n = hip_lib().xpnghip_shard_ranges(w, h, D, arr, max(D, 1))
    if n < 0:
        raise XpngError("xpnghip_shard_ranges failed")
    return [(int(arr[2 * k]), int(arr[2 * k + 1])) for k in range(n)]


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise XpngError(f"{HOST_SO} is missing: run `make host`")
        hip_lib()
        L = C.CDLL(HOST_SO)
        for name in ("xpng_store", "xpng_load", "store_7", "load_7", "xpng_store_T", "xpng_load_T"):
            getattr(L, name).restype = C.c_bool
        L.xpng_store.argtypes = [C.c_uint64, C.POINTER(XpngT), C.c_char_p]
        L.xpng_load.argtypes = [C.c_char_p, C.POINTER(XpngT)]
        L.xpng_store_T.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(XpngT), C.c_char_p]
        L.xpng_load_T.argtypes = [C.c_uint64, C.c_char_p, C.POINTER(XpngT)]
        L.store_7.argtypes = [C.POINTER(XpngT), C.c_char_p]
        L.load_7.argtypes = [C.c_char_p, C.POINTER(XpngT)]
        _host = L
    return _host


def _err():
    return hip_lib().xpnghip_last_error().decode(errors="replace")


def device_count() -> int:
    return hip_lib().xpnghip_device_count()


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class MallocedBlobs:
    """The malloc()ed buffer xpnghip_encode_tiles hands back, not yet copied into a Python object (bench.py times the C call
    alone: the copy into `bytes` and the free() are this binding's, not the library's)."""
    def __init__(self, p, n):
        self.p, self.n = p, n

    def bytes(self) -> bytes:
        return C.string_at(self.p, self.n)

    def free(self):
        if self.p:
            _libc.free(self.p)
            self.p = None


def encode_tiles(mode: int, raster: np.ndarray, T: int = 1, lib=None, copy: bool = True):
    """Host raster (h, w, 3|4) uint8 -> concatenated tile blobs (xpnghip_encode_tiles_T; H2D + kernels + D2H) on T devices.
    copy=False returns the library's malloc()ed buffer as a MallocedBlobs (caller frees)."""
    raster = np.ascontiguousarray(raster, dtype=np.uint8)
    h, w, ch = raster.shape
    p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
    lib = lib or hip_lib()
    if lib.xpnghip_encode_tiles_T(T, mode, raster.ctypes.data_as(C.c_void_p), w, h, ch, C.byref(p), C.byref(n)):
        raise XpngError("xpnghip_encode_tiles: " + lib.xpnghip_last_error().decode(errors="replace"))
    if not copy:
        return MallocedBlobs(p, n.value)
    out = C.string_at(p, n.value)
    _libc.free(p)
    return out


def decode_tiles(mode: int, blobs: bytes, w: int, h: int, pxsz: int, T: int = 1, lib=None, out: np.ndarray = None) -> np.ndarray:
    """Tile blobs -> raster (xpnghip_decode_tiles_T).  out: a caller-allocated (h, w, pxsz) uint8 array to fill (what xpng_load's
    malloc is to the C call, libxpng.c:974); default: a fresh zero-filled one."""
    raster = np.zeros((h, w, pxsz), dtype=np.uint8) if out is None else out
    assert raster.shape == (h, w, pxsz) and raster.dtype == np.uint8 and raster.flags["C_CONTIGUOUS"]
    buf = np.frombuffer(blobs, dtype=np.uint8)
    lib = lib or hip_lib()
    if lib.xpnghip_decode_tiles_T(T, mode, buf.ctypes.data_as(C.c_void_p), len(blobs), w, h, pxsz,
                                  raster.ctypes.data_as(C.c_void_p)):
        raise XpngError("xpnghip_decode_tiles: " + lib.xpnghip_last_error().decode(errors="replace"))
    return raster


def staged_encode(mode: int, raster: np.ndarray, T: int = 1, lib=None):
    """The staged-image sequence xpng_store runs (xpnghip_image_begin .. _end): returns (bytes per pixel after
    normalize_RGBA, single-colour?, tile blobs, normalised raster bytes)."""
    raster = np.ascontiguousarray(raster, dtype=np.uint8)
    h, w, ch = raster.shape
    lib = lib or hip_lib()
    img, pxsz = C.c_void_p(), C.c_int(0)
    if lib.xpnghip_image_begin(C.byref(img), raster.ctypes.data_as(C.c_void_p), w, h, ch, C.byref(pxsz)):
        raise XpngError("xpnghip_image_begin: " + lib.xpnghip_last_error().decode(errors="replace"))
    try:
        single = C.c_int(0)
        if lib.xpnghip_image_single_colour(img, C.byref(single)):
            raise XpngError("xpnghip_image_single_colour: " + lib.xpnghip_last_error().decode(errors="replace"))
        norm = np.zeros(h * w * pxsz.value, dtype=np.uint8)
        if lib.xpnghip_image_fetch(img, norm.ctypes.data_as(C.c_void_p)):
            raise XpngError("xpnghip_image_fetch: " + lib.xpnghip_last_error().decode(errors="replace"))
        p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        if lib.xpnghip_image_encode_T(img, T, mode, C.byref(p), C.byref(n)):
            raise XpngError("xpnghip_image_encode_T: " + lib.xpnghip_last_error().decode(errors="replace"))
        blobs = C.string_at(p, n.value)
        _libc.free(p)
        return pxsz.value, bool(single.value), blobs, norm.tobytes()
    finally:
        lib.xpnghip_image_end(img)


def image_store(mode: int, raster: np.ndarray, T: int = 0, lib=None) -> MallocedBlobs:
    """Exactly the device-side sequence of xpng_store for a raster that reaches the tile codec (host/xpng_api.c store_on_device):
    xpnghip_image_begin -> xpnghip_image_encode_T -> xpnghip_image_end.  Returns the library's malloc()ed blob buffer."""
    raster = np.ascontiguousarray(raster, dtype=np.uint8)
    h, w, ch = raster.shape
    lib = lib or hip_lib()
    img, pxsz = C.c_void_p(), C.c_int(0)
    if lib.xpnghip_image_begin(C.byref(img), raster.ctypes.data_as(C.c_void_p), w, h, ch, C.byref(pxsz)):
        raise XpngError("xpnghip_image_begin: " + lib.xpnghip_last_error().decode(errors="replace"))
    try:
        p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        if lib.xpnghip_image_encode_T(img, T, mode, C.byref(p), C.byref(n)):
            raise XpngError("xpnghip_image_encode_T: " + lib.xpnghip_last_error().decode(errors="replace"))
        return MallocedBlobs(p, n.value)
    finally:
        lib.xpnghip_image_end(img)


def store(mode: int, raster: np.ndarray, path: str, T: int = None) -> None:
    """xpng_store[_T] (include/xpng.h): full host driver incl. normalisation, fallbacks and file output."""
    raster = np.ascontiguousarray(raster, dtype=np.uint8)
    h, w, ch = raster.shape
    pm = XpngT(raster.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, raster.size, ch == 4)
    rc = host_lib().xpng_store(mode, C.byref(pm), path.encode()) if T is None else host_lib().xpng_store_T(T, mode, C.byref(pm), path.encode())
    if rc:
        raise XpngError("xpng_store failed")


def load(path: str, T: int = None) -> np.ndarray:
    pm = XpngT()
    rc = host_lib().xpng_load(path.encode(), C.byref(pm)) if T is None else host_lib().xpng_load_T(T, path.encode(), C.byref(pm))
    if rc:
        raise XpngError("xpng_load failed")
    out = np.ctypeslib.as_array(pm.p, shape=(pm.h, pm.w, 3 + int(pm.A))).copy()
    _libc.free(pm.p)
    return out


def normalize_device(d_rgba: int, npx: int, d_out: int, stream=0):
    """normalize_RGBA (libxpng.c:688-721) on a device-resident RGBA raster -> (bytes per pixel, rewritten into d_out?)."""
    pxsz, rew = C.c_int(0), C.c_int(0)
    if hip_lib().xpnghip_normalize_device(d_rgba, npx, d_out, C.byref(pxsz), C.byref(rew), stream):
        raise XpngError("xpnghip_normalize_device: " + _err())
    return pxsz.value, bool(rew.value)


class Context:
    """xpnghip_ctx: tile table + device workspace for one raster geometry on one GPU.  Device pointers are
    plain integers (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t handle or 0."""

    FETCH = {"pr": 0, "nl": 1, "r": 2, "g": 3, "b": 4, "a": 5, "k": 19, "sums": 30}

    def __init__(self, w: int, h: int, pxsz: int, device: int = 0, batch: int = 1, tile_range=None):
        self.w, self.h, self.pxsz, self.device, self.batch = w, h, pxsz, device, batch
        self._h = C.c_void_p()
        r0, r1 = tile_range if tile_range else (0, (1 << 64) - 1)
        if hip_lib().xpnghip_ctx_create_range(C.byref(self._h), device, w, h, pxsz, batch, r0, r1):
            raise XpngError("xpnghip_ctx_create_range: " + _err())
        self.n_tiles = hip_lib().xpnghip_ctx_tile_count(self._h)

    def close(self):
        if self._h:
            hip_lib().xpnghip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tile(self, i: int):
        a = (C.c_uint64 * 4)()
        if hip_lib().xpnghip_ctx_tile(self._h, i, a):
            raise IndexError(i)
        return tuple(a)

    def tiles(self):
        return [self.tile(i) for i in range(self.n_tiles)]

    def blob_bound(self, t0=0, t1=None) -> int:
        return hip_lib().xpnghip_ctx_blob_bound(self._h, t0, self.n_tiles if t1 is None else t1)

    def workspace_bytes(self) -> int:
        return hip_lib().xpnghip_ctx_workspace_bytes(self._h)

    def encode_device(self, mode, d_raster: int, d_blobs: int, t0=0, t1=None, stream=0, sync=True) -> int:
        n = C.c_uint64()
        rc = hip_lib().xpnghip_encode_device(self._h, mode, d_raster, t0, self.n_tiles if t1 is None else t1, d_blobs,
                                             C.byref(n) if sync else None, stream)
        if rc:
            raise XpngError("xpnghip_encode_device: " + _err())
        return n.value

    def encode_device_batch(self, mode, d_rasters, d_blobs, t0=0, t1=None, stream=0, sync=True):
        """d_rasters / d_blobs: sequences of device pointers (one per image, <= batch).  Returns the list of blob lengths
        (sync=True) or None."""
        k = len(d_rasters)
        ins, outs, lens = (C.c_void_p * k)(*d_rasters), (C.c_void_p * k)(*d_blobs), (C.c_uint64 * k)()
        rc = hip_lib().xpnghip_encode_device_batch(self._h, mode, ins, k, t0, self.n_tiles if t1 is None else t1, outs,
                                                   lens if sync else None, stream)
        if rc:
            raise XpngError("xpnghip_encode_device_batch: " + _err())
        return list(lens) if sync else None

    def decode_status(self, stream=0) -> int:
        """Synchronise and report whether the last decode accepted every tile header (0) or rejected some (1)."""
        return hip_lib().xpnghip_ctx_decode_status(self._h, stream)

    def decode_device_batch(self, mode, d_blobs, blob_lens, tile_offs, d_rasters, t0=0, t1=None, stream=0):
        """blob_lens: bytes of each blob buffer; tile_offs: per image, the blob start offsets of tiles [t0, t1)."""
        k = len(d_blobs)
        t1 = self.n_tiles if t1 is None else t1
        off_arr = None   # tile_offs None: the size walk (libxpng.c:982) runs on the device
        if tile_offs is not None:
            flat = [o for offs in tile_offs for o in offs]
            assert len(flat) == k * (t1 - t0)
            key = (tuple(flat), t0, t1)
            if getattr(self, "_off_key", None) != key:
                self._off_key, self._off_arr = key, (C.c_uint64 * len(flat))(*flat)
            off_arr = self._off_arr
        ins, outs, lens = (C.c_void_p * k)(*d_blobs), (C.c_void_p * k)(*d_rasters), (C.c_uint64 * k)(*blob_lens)
        if hip_lib().xpnghip_decode_device_batch(self._h, mode, ins, lens, k, off_arr, t0, t1, outs, stream):
            raise XpngError("xpnghip_decode_device_batch: " + _err())

    def last_blobs_len(self) -> int:
        return hip_lib().xpnghip_ctx_last_blobs_len(self._h)

    def decode_device(self, mode, d_blobs: int, blobs_len: int, tile_off, d_raster: int, t0=0, t1=None, stream=0):
        t1 = self.n_tiles if t1 is None else t1
        arr = (C.c_uint64 * (t1 - t0))(*tile_off) if tile_off is not None else None
        if hip_lib().xpnghip_decode_device(self._h, mode, d_blobs, blobs_len, arr, t0, t1, d_raster, stream):
            raise XpngError("xpnghip_decode_device: " + _err())

    def transform_device(self, d_raster: int, t0=0, t1=None, stream=0):
        if hip_lib().xpnghip_m1_transform_device(self._h, d_raster, t0, self.n_tiles if t1 is None else t1, stream):
            raise XpngError("xpnghip_m1_transform_device: " + _err())

    def transform_device_batch(self, d_rasters, t0=0, t1=None, stream=0):
        k = len(d_rasters)
        ins = (C.c_void_p * k)(*d_rasters)
        if hip_lib().xpnghip_m1_transform_device_batch(self._h, ins, k, t0, self.n_tiles if t1 is None else t1, stream):
            raise XpngError("xpnghip_m1_transform_device_batch: " + _err())

    def fetch(self, what, tile: int, cap: int = 1 << 22) -> np.ndarray:
        code = self.FETCH[what] if isinstance(what, str) else what
        buf = np.zeros(cap, dtype=np.uint8)
        n = hip_lib().xpnghip_debug_fetch(self._h, code, tile, buf.ctypes.data_as(C.c_void_p), cap)
        if n < 0:
            raise XpngError(f"debug_fetch({what}, {tile}) failed")
        return buf[:n].copy()


def walk_tile_offsets(blobs: bytes, n_tiles: int):
    """Serial size walk of the reference decoder (libxpng.c:982): blob start offsets of every tile."""
    off, o = [], 0
    for _ in range(n_tiles):
        off.append(o)
        o += int.from_bytes(blobs[o:o + 4], "little") & 0xFFFFFF
    return off, o
