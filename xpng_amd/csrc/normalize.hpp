// normalize.hpp -- normalize_RGBA (libxpng.c:688-721) and the whole-image single-colour test (libxpng.c:741-753) on the
// device, so that xpng_store uploads the caller's raster once and never walks it on the host.
//
//   hidden      = exists px: alpha == 0 and (r | g | b) != 0   -> rewrite: those pixels become 0 (stays RGBA)
//   translucent = exists px: alpha != 255                      -> else if none: repack to RGB
// (the reference's early-break loop is equivalent to these two ORs: SURVEY.md §8(a) row E)
#pragma once
#include "common.hpp"

namespace xpng {

// flags[0] |= hidden, flags[1] |= translucent.  One uint4 (4 pixels) per thread per iteration; px needs only its natural 4-byte
// alignment (a band of a raster starts at any pixel): the pixels in front of the first 16-byte boundary and behind the last whole
// group are looked at one by one.
__global__ __launch_bounds__(256) void k_norm_flags(const uint32_t *__restrict__ px, uint64_t n, uint32_t *__restrict__ flags) {
    uint32_t hidden = 0, transl = 0;
    const uint64_t head = min(n, (uint64_t)(((16u - (uint32_t)((uintptr_t)px & 15u)) & 15u) >> 2));
    const uint32_t *body = px + head;
    const uint64_t nb = n - head, stride = (uint64_t)gridDim.x * blockDim.x, n4 = nb / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(body)[i];
        const uint32_t p[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t a = p[k] >> 24;
            hidden |= (a == 0 && (p[k] & 0xFFFFFFu)) ? 1u : 0u;
            transl |= a != 255 ? 1u : 0u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {  // head pixels (threads 0..2) and tail pixels (threads 4..6)
        const uint32_t t = threadIdx.x & 3u;
        const bool tail = threadIdx.x >= 4;
        if (t < (tail ? (nb & 3) : head)) {
            const uint32_t v = tail ? body[n4 * 4 + t] : px[t], a = v >> 24;
            hidden |= (a == 0 && (v & 0xFFFFFFu)) ? 1u : 0u;
            transl |= a != 255 ? 1u : 0u;
        }
    }
    if (__ballot(hidden) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1u);
    if (__ballot(transl) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1u);
}

// the rewrite of the "hidden" case IN PLACE on a raster this library owns, and only when flags[0] says there is something to
// rewrite (the kernel is launched behind k_norm_flags on the same stream without the host looking at the flag in between): pixels
// with alpha 0 become 0 (libxpng.c:699-707), everything else is left alone - nothing is written for a pixel that does not change
__global__ __launch_bounds__(256) void k_norm_zero_hidden_if(const uint32_t *__restrict__ flags, uint32_t *__restrict__ px, uint64_t n) {
    if (!flags[0]) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = px[i];
        if ((v >> 24) == 0 && v) px[i] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_norm_zero_hidden(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = in[i];
        out[i] = (v >> 24) ? v : 0u;
    }
}

// RGBA -> RGB: 4 pixels (one uint4) in, 3 dwords out per thread
__global__ __launch_bounds__(256) void k_norm_to_rgb(const uint32_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        const uint32_t a = v.x & 0xFFFFFFu, b = v.y & 0xFFFFFFu, c = v.z & 0xFFFFFFu, d = v.w & 0xFFFFFFu;
        uint32_t *o = reinterpret_cast<uint32_t *>(out) + 3 * i;
        o[0] = a | (b << 24); o[1] = (b >> 8) | (c << 16); o[2] = (c >> 16) | (d << 8);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        const uint32_t v = in[i];
        out[3 * i] = (uint8_t)v; out[3 * i + 1] = (uint8_t)(v >> 8); out[3 * i + 2] = (uint8_t)(v >> 16);
    }
}

// flag[0] |= 1 if some pixel differs from pixel 0
template <int PXSZ>
__global__ __launch_bounds__(256) void k_any_differs(const uint8_t *__restrict__ p, uint64_t n, uint32_t *__restrict__ flag) {
    const uint32_t first = load_px<PXSZ>(p) & (PXSZ == 4 ? 0xFFFFFFFFu : 0xFFFFFFu);
    uint32_t diff = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t v;
        if (PXSZ == 4) v = reinterpret_cast<const uint32_t *>(p)[i];
        else v = (uint32_t)p[3 * i] | ((uint32_t)p[3 * i + 1] << 8) | ((uint32_t)p[3 * i + 2] << 16);
        diff |= v != first ? 1u : 0u;
    }
    if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

}  // namespace xpng
