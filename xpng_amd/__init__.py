"""xpng_amd -- MI355X-native xPNG tile codec (host-side Python mirror of the C API).

The product is native: `xpng_amd/lib/libxpng_hip.so` (hand-written HIP kernels for gfx950 behind the C-ABI
of include/xpng_hip.h) and `libxpng.so` / `bin/xpng` (the host C driver and CLI that mirror the reference's
xpng.h / xpng.c).  This package only binds those libraries with ctypes for tests, bench.py and
torch.distributed sharding.  It never falls back to a CPU codec: if the HIP library is missing or no GPU
is visible, compute calls raise.
"""
from .api import (Context, XpngError, build_native, decode_tiles, device_count, encode_tiles, hip_lib,
                  host_lib, load, native_paths, normalize_device, store)

__all__ = ["Context", "XpngError", "build_native", "decode_tiles", "device_count", "encode_tiles", "hip_lib",
           "host_lib", "load", "native_paths", "normalize_device", "store"]
