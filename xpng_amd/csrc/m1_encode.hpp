// m1_encode.hpp -- mode-1 (XPNG_COMPRESSION_TYPE_FAST) tile ENCODE kernels for gfx950.
//
// Reference path restated: enc_1_th (libxpng.c:534-571) = pp_rgbx (92-140) + m1e_* (497-530) +
// compress_block_v2 (307-427) + tile container.  The reference fuses everything into one serial loop per
// tile; here the same bytes come out of five data-parallel stages:
//
//   k_chooser        sampled 4-way cost sums per tile            (pp_rgbx)
//   k_m1_transform   per-pixel residual / zig-zag / nl / alpha   (M1ENC arithmetic) -> 5 symbol planes
//   k_m1_streams     routing: 9 context streams + bit stream k   (pl chain, BITSTREAM_WRITE) via wave scans
//   k_rans2_encode   one wavefront per (tile, stream): histogram, normalise, tables, 2-lane rANS, splice
//   k_tile_sizes / k_tile_offsets / k_tile_gather                 tile blob container + concatenation
#pragma once
#include "common.hpp"

namespace xpng {

// --------------------------------------------------------------------------------------------------
// K1  predictor chooser: reference pp_rgbx, libxpng.c:92-140.  Every 4th pixel in x and y (x%4==3,
// y%4==3) is costed under avg / avg+G / grad / grad+G; the four sums go to sums[tile*4..] by atomics.
// grid = tiles * strips, block = 256.
template <int PXSZ>
__global__ __launch_bounds__(256) void k_chooser(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                 const TileDesc *__restrict__ tiles, TileSel sel, uint32_t strips,
                                                 uint32_t *__restrict__ sums) {
    bw_prio();
    const uint32_t tile = vtile(sel, blockIdx.x / strips), strip = blockIdx.x % strips;
    const TileDesc t = tiles[tile];
    const uint8_t *__restrict__ raster = rasters[t.img];
    if (t.w < 4 || t.h < 4) return;
    const uint32_t xs = t.w >> 2, ys = t.h >> 2;
    const uint32_t j0 = (uint32_t)((uint64_t)ys * strip / strips), j1 = (uint32_t)((uint64_t)ys * (strip + 1) / strips);
    const uint32_t cnt = (j1 - j0) * xs;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const bool pair_loads = PXSZ == 4 && (t.x & 1u) == 0 && (bpr & 7u) == 0;  // pixel x-1 (x % 4 == 3) is then 8-byte aligned: L|cur and UL|U in one load each
    for (uint32_t idx = threadIdx.x; idx < cnt; idx += 256) {
        const uint32_t j = j0 + idx / xs, i = idx - (idx / xs) * xs;
        const uint8_t *p = raster + (uint64_t)(t.y + 4 * j + 3) * bpr + (uint64_t)(t.x + 4 * i + 3) * PXSZ;
        uint32_t cur, L, U, UL;
        if (pair_loads) {
            const uint2 a = *reinterpret_cast<const uint2 *>(p - 4), b = *reinterpret_cast<const uint2 *>(p - bpr - 4);
            L = a.x; cur = a.y; UL = b.x; U = b.y;
        } else if (PXSZ == 3) {
            // left + current pixel = 6 consecutive bytes: one unaligned 8-byte load that ENDS with the current pixel (so it
            // never leaves the raster; x >= 3, so it does not start in front of it either) instead of six byte loads
            typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
            const u32x2_a1 a = *reinterpret_cast<const u32x2_a1 *>(p - 5), b = *reinterpret_cast<const u32x2_a1 *>(p - bpr - 5);
            L = __builtin_amdgcn_alignbyte(a.y, a.x, 2) & 0xFFFFFFu; cur = a.y >> 8;
            UL = __builtin_amdgcn_alignbyte(b.y, b.x, 2) & 0xFFFFFFu; U = b.y >> 8;
        } else {
            cur = load_px<PXSZ>(p); L = load_px<PXSZ>(p - PXSZ); U = load_px<PXSZ>(p - bpr); UL = load_px<PXSZ>(p - bpr - PXSZ);
        }
        if (PXSZ == 4 && (cur >> 24) == 0) continue;  // libxpng.c:121
        int d2[3], d3[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v = (cur >> (8 * c)) & 255, l = (L >> (8 * c)) & 255, u = (U >> (8 * c)) & 255, ul = (UL >> (8 * c)) & 255;
            d2[c] = v - pred_avg(l, u);
            d3[c] = v - pred_grad(l, u, ul);
        }
        c0 += bit_width(zz_enc(d2[0]) | zz_enc(d2[1]) | zz_enc(d2[2]));
        c1 += bit_width(zz_enc(d2[0] - d2[1]) | zz_enc(d2[1]) | zz_enc(d2[2] - d2[1]));
        c2 += bit_width(zz_enc(d3[0]) | zz_enc(d3[1]) | zz_enc(d3[2]));
        c3 += bit_width(zz_enc(d3[0] - d3[1]) | zz_enc(d3[1]) | zz_enc(d3[2] - d3[1]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off);
        c1 += __shfl_down(c1, off);
        c2 += __shfl_down(c2, off);
        c3 += __shfl_down(c3, off);
    }
    if ((threadIdx.x & 63) == 0) {
        uint32_t *s = sums + (uint64_t)tile * 4;
        if (c0) atomicAdd(s + 0, c0);
        if (c1) atomicAdd(s + 1, c1);
        if (c2) atomicAdd(s + 2, c2);
        if (c3) atomicAdd(s + 3, c3);
    }
}

// --------------------------------------------------------------------------------------------------
// Per-pixel arithmetic of M1ENC / M1ENC4 (libxpng.c:497-513) without the routing.  cur/L/U/UL are packed
// pixels (r | g<<8 | b<<16 | a<<24).  Outputs: nl (or NL_NONE), zig-zag residual bytes, alpha symbol.
template <int PXSZ>
__device__ __forceinline__ void m1_pixel(uint32_t cur, uint32_t L, uint32_t U, uint32_t UL, bool row0, bool col0,
                                         int useGrad, int useG, uint32_t &nl, uint32_t &zr, uint32_t &zg, uint32_t &zb,
                                         uint32_t &za) {
    nl = NL_NONE; zr = zg = zb = za = 0;
    if constexpr (PXSZ == 4) {
        const int ca = cur >> 24;
        const int pa = (row0 || !col0) ? (int)(L >> 24) : (int)(U >> 24);  // libxpng.c:510-511: left, except in column 0
        za = (uint32_t)zz_enc(ca - pa);
        if (ca == 0) return;  // libxpng.c:502: invisible pixel emits alpha only
    }
    int d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int v = (cur >> (8 * c)) & 255, l = (L >> (8 * c)) & 255, u = (U >> (8 * c)) & 255, ul = (UL >> (8 * c)) & 255;
        int pred;
        if (row0) pred = l;
        else if (col0) pred = u;
        else pred = useGrad ? pred_grad(l, u, ul) : pred_avg(l, u);
        d[c] = v - pred;
    }
    if (useG && !row0 && !col0) { d[0] -= d[1]; d[2] -= d[1]; }  // libxpng.c:513
    zr = (uint32_t)zz_enc(d[0]); zg = (uint32_t)zz_enc(d[1]); zb = (uint32_t)zz_enc(d[2]);
    nl = (uint32_t)bit_width(zr | zg | zb);
}

// --------------------------------------------------------------------------------------------------
// K2 (generic form)  per-pixel transform straight from global memory; one thread = 4 consecutive pixels of
// the tile in raster order, so every plane store is one aligned dword.  Works for any tile geometry
// (including the very wide / very tall tiles of images narrower than 444 px); the LDS-staged fast form
// below takes over for ordinary tiles.
// grid = tiles * blocks_per_tile, block = 256 (1024 pixels per block).
// ---- byte-parallel (SWAR) forms of the per-pixel arithmetic for interior RGBA pixels: all four channels of a pixel
// move through one 32-bit register (r | g<<8 | b<<16 | a<<24).
__device__ __forceinline__ uint32_t swar_sub8(uint32_t a, uint32_t b) {  // per-byte a - b (mod 256)
    return ((a | 0x80808080u) - (b & 0x7F7F7F7Fu)) ^ ((a ^ ~b) & 0x80808080u);
}
__device__ __forceinline__ uint32_t swar_zigzag8(uint32_t d) {  // per byte: v = (int8)d; (v << 1) ^ (v >> 7)   (pix_toU)
    const uint32_t sgn = (d >> 7) & 0x01010101u;
    return ((d << 1) & 0xFEFEFEFEu) ^ ((sgn << 8) - sgn);
}
// interior pixel (row > 0, column > 0): returns the zig-zag word (zr | zg<<8 | zb<<16 | za<<24) and nl.
// col0: the pixel is in column 0 of its tile (row > 0): every channel predicts from U and the green subtraction is skipped
// (libxpng.c:505-513; it applies to interior pixels only).
__device__ __forceinline__ uint32_t m1_pixel_interior(const int useGrad, const int useG, uint32_t cur, uint32_t L, uint32_t U, uint32_t UL, uint32_t &nl, bool col0 = false) {
    uint32_t pred;
    if (!useGrad) {
        pred = __builtin_amdgcn_lerp(L, U, 0x01010101u);  // per byte (L + U + 1) >> 1   (p2a)
    } else {
        // p3a = ((3L + 3U - 2UL) + 2) >> 2 (arithmetic), low 8 bits.  r and b together as two 16-bit lanes: adding 1024 keeps
        // every lane positive (range -508 .. 1532) and, being a multiple of 4 * 256, leaves the low 8 bits of the shifted
        // value unchanged; g alone in a 32-bit register with a plain arithmetic shift.
        const uint32_t Le = L & 0x00FF00FFu, Ue = U & 0x00FF00FFu, ULe = UL & 0x00FF00FFu;
        const uint32_t te = times3(Le + Ue) + 0x04020402u - 2u * ULe;
        const uint32_t pe = (te >> 2) & 0x00FF00FFu;
        const int lg = (int)((L >> 8) & 255u), ug = (int)((U >> 8) & 255u), ulg = (int)((UL >> 8) & 255u);
        const uint32_t pg = (uint32_t)(((int)times3((uint32_t)(lg + ug)) - 2 * ulg + 2) >> 2) & 255u;
        pred = pe | (pg << 8);
    }
    pred = (pred & 0x00FFFFFFu) | (L & 0xFF000000u);  // alpha always predicts from the left (libxpng.c:511)
    pred = col0 ? U : pred;
    uint32_t d = swar_sub8(cur, pred);
    if (useG) {                                        // r -= g, b -= g on the residuals (libxpng.c:513)
        const uint32_t g = (d >> 8) & 0xFFu;
        const uint32_t dg = swar_sub8(d, g | (g << 16));
        d = col0 ? d : dg;
    }
    uint32_t z = swar_zigzag8(d);
    const uint32_t m = (z | (z >> 8) | (z >> 16)) & 0xFFu;
    nl = (uint32_t)bit_width(m);
    if ((cur >> 24) == 0) { nl = NL_NONE; z &= 0xFF000000u; }  // invisible pixel: alpha symbol only (libxpng.c:502)
    return z;
}

// four interior RGB pixels (left neighbour cp[0], pixels cp[1..4]; the five above them in up[]) through the byte-parallel
// arithmetic: the alpha byte is set to 255 on both sides, so it predicts itself and leaves a zero residual
__device__ __forceinline__ void rgb_group_interior(const int useGrad, const int useG, const uint32_t *cp, const uint32_t *up, uint32_t &onl, uint32_t &orr, uint32_t &og, uint32_t &ob) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t nl;
        const uint32_t z = m1_pixel_interior(useGrad, useG, cp[k + 1] | 0xFF000000u, cp[k] | 0xFF000000u, up[k + 1], up[k], nl);
        onl |= nl << (8 * k); orr |= (z & 255u) << (8 * k); og |= ((z >> 8) & 255u) << (8 * k); ob |= ((z >> 16) & 255u) << (8 * k);
    }
}


// ---- stream lengths without a pass of their own [r4] ----------------------------------------------------------------------------
// The nine context-stream lengths of a tile are a histogram of its nl plane (k_m1_lens below).  Through round 3 a kernel of its own
// (k_m1_count) re-read the plane for it: 1 B/px of HBM traffic and a launch on the encode's critical path.  The transform holds every
// nl it writes, so it counts them itself: a thread keeps packed per-bin counters of its own pixels (NlAcc below), remembers its last
// group with a coded pixel, and the workgroup leaves its nine sums and ((index of its last coded pixel + 1) << 4 | that pixel's nl) in
// a record of its own: nlh[(tile * slots + workgroup of the tile) * NLH_STRIDE + 0..9], plain stores.  (A first version added them to
// per-tile counters with ten atomics per workgroup: 7 M memory-side atomics per 128-raster launch, ~4 % of the transform's time.)
// k_m1_lens / k_m2_count add a tile's records up (nlh_reduce).
constexpr uint32_t NLH_STRIDE = 12;  // u32 per tile: nine counts, the last-coded-pixel key, two spare
#ifndef XPNG_HIST_VARIANT
#define XPNG_HIST_VARIANT 1
#endif
#if XPNG_HIST_VARIANT == 0
// one 64-bit accumulator of nine 6-bit fields (a thread sees at most 24 pixels of a strip; the marker NL_NONE = 255 lands at bit
// (6 * 255) & 63 = 58, above the nine fields, and whatever it carries out of bit 63 is gone)
struct NlAcc { uint64_t acc; uint32_t last_onl, last_i0; };
__device__ __forceinline__ NlAcc nlacc_zero() { return NlAcc{0ull, 0xFFFFFFFFu, 0u}; }
// onl = the nl bytes of four consecutive pixels (NL_NONE: not coded), i0 = index of the first of them inside the tile
__device__ __forceinline__ void nlacc_add(NlAcc &a, uint32_t onl, uint32_t i0) {
#pragma unroll
    for (int k = 0; k < 4; k++) a.acc += 1ull << ((6u * ((onl >> (8 * k)) & 255u)) & 63u);
    const bool any = onl != 0xFFFFFFFFu;
    a.last_onl = any ? onl : a.last_onl;  // (a thread's groups come in increasing pixel order: the last assignment wins)
    a.last_i0 = any ? i0 : a.last_i0;
}
__device__ __forceinline__ void nlacc_fields(const NlAcc &a, uint32_t *f) {
#pragma unroll
    for (int c = 0; c < 9; c++) f[c] = (uint32_t)(a.acc >> (6 * c)) & 63u;
}
#else
// four 32-bit accumulators, one per pixel position of a group, each of nine 3-bit fields: a thread sees at most six groups of a
// strip (seven would still fit), so a field counts to 6; one pixel is a multiply-by-3, a shift and an add.  The marker NL_NONE = 255
// lands at bit (3 * 255) & 31 = 29, above the nine fields (bits 0..26), and whatever it carries out of bit 31 is gone.
struct NlAcc { uint32_t acc[4]; uint32_t last_onl, last_i0; };
__device__ __forceinline__ NlAcc nlacc_zero() { return NlAcc{{0u, 0u, 0u, 0u}, 0xFFFFFFFFu, 0u}; }
// onl = the nl bytes of four consecutive pixels (NL_NONE: not coded), i0 = index of the first of them inside the tile
__device__ __forceinline__ void nlacc_add(NlAcc &a, uint32_t onl, uint32_t i0) {
#pragma unroll
    for (int k = 0; k < 4; k++) a.acc[k] += 1u << (__umul24((onl >> (8 * k)) & 255u, 3u) & 31u);
    const bool any = onl != 0xFFFFFFFFu;
    a.last_onl = any ? onl : a.last_onl;  // (a thread's groups come in increasing pixel order: the last assignment wins)
    a.last_i0 = any ? i0 : a.last_i0;
}
__device__ __forceinline__ void nlacc_fields(const NlAcc &a, uint32_t *f) {
    // even fields (bins 0, 2, 4, 6, 8) and odd fields (1, 3, 5, 7) apart, so that the four accumulators add up in 6-bit fields
    uint32_t ev = 0, od = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { ev += a.acc[k] & 0x71C71C7u; od += (a.acc[k] >> 3) & 0x1C71C7u; }
#pragma unroll
    for (int c = 0; c < 9; c++) f[c] = (((c & 1) ? od : ev) >> (6 * (c >> 1))) & 63u;
}
#endif
// the workgroup's sums -> the tile's counters.  Every thread of the THREADS-thread workgroup calls it once (it holds a barrier);
// s_red: THREADS / 64 x 6 words of LDS nobody else uses.
template <int THREADS>
__device__ __forceinline__ void nlacc_commit(const NlAcc &a, uint32_t (*s_red)[6], uint32_t *__restrict__ nlh_tile) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t f[9];
    nlacc_fields(a, f);
    // two 16-bit fields per word: a workgroup sees at most 5376 pixels
    uint32_t w[5] = {f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16), f[8]};
    uint32_t key = 0;
    if (a.last_onl != 0xFFFFFFFFu) {
        const uint32_t kb = (31u - (uint32_t)__clz((int)~a.last_onl)) >> 3;  // highest byte that is not NL_NONE (its complement is not 0)
        key = ((a.last_i0 + kb + 1u) << 4) | ((a.last_onl >> (8 * kb)) & 15u);
    }
#pragma unroll
    for (int q = 0; q < 5; q++) w[q] = wave_scan_incl(w[q]);
    key = wave_scan_max(key);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 5; q++) s_red[wv][q] = w[q];
        s_red[wv][5] = key;
    }
    __syncthreads();
    if (tid < 10) {
        uint32_t v = 0;
#pragma unroll
        for (int w2 = 0; w2 < THREADS / 64; w2++) {
            const uint32_t x = s_red[w2][tid < 9 ? tid >> 1 : 5];
            if (tid < 9) v += (tid & 1) ? x >> 16 : x & 0xFFFFu;
            else v = x > v ? x : v;
        }
        nlh_tile[tid] = v;  // (this workgroup's own record: every record a launch's reduction reads is written by that launch)
    }
}
// records of a tile -> h[0..8] = histogram of its nl plane over coded pixels, h[9] = key of its last coded pixel (0: none), the same in
// every lane.  One wavefront; nrec = records the transform wrote for this tile (its strips, or its 4096-pixel chunks).
__device__ __forceinline__ void nlh_reduce(const uint32_t *__restrict__ rec, uint32_t nrec, uint32_t *h) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t r = lane; r < nrec; r += 64) {
        const uint4 *p = reinterpret_cast<const uint4 *>(rec + (uint64_t)r * NLH_STRIDE);  // (records are 48 bytes: 16-byte aligned)
        const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
        a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w; a[8] += v2.x;
        a[9] = v2.y > a[9] ? v2.y : a[9];
    }
#pragma unroll
    for (int c = 0; c < 9; c++) h[c] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(a[c]), 63);
    h[9] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max(a[9]), 63);
}
// how many records the transform left for a tile: one per strip of TR_ROWS rows (LDS-staged forms) or per 1024 * TG_REPS pixels
__host__ __device__ inline uint32_t nlh_records(uint32_t w, uint32_t h, bool generic);
// records -> the nine stream lengths (common.hpp: stream-scratch layout): stream c receives the nl of every coded pixel whose
// predecessor (the previous coded pixel; the first one's is 0, libxpng.c:497-508) has nl = c:
//     len[c] = hist[c] - [c == nl of the last coded pixel] + [c == 0]            (all zero when the tile codes no pixel)
// grid = tiles, block = 64.
__global__ __launch_bounds__(64) void k_m1_lens(const uint32_t *__restrict__ nlh, uint32_t slots, uint32_t generic, const TileDesc *__restrict__ tiles,
                                                TileSel sel, uint32_t *__restrict__ ctx_n) {
    bw_prio();
    const uint32_t tile = vtile(sel, blockIdx.x), lane = threadIdx.x & 63;
    const TileDesc t = tiles[tile];
    uint32_t h[10];
    nlh_reduce(nlh + (uint64_t)tile * slots * NLH_STRIDE, nlh_records(t.w, t.h, generic != 0), h);
    if (lane < 9) {
        uint32_t len = 0;
#pragma unroll
        for (int c = 0; c < 9; c++) len = lane == (uint32_t)c ? h[c] : len;
        if (h[9]) len = len - ((h[9] & 15u) == lane ? 1u : 0u) + (lane == 0 ? 1u : 0u);
        ctx_n[(uint64_t)tile * 9 + lane] = len;
    }
}

constexpr uint32_t TG_REPS = 4;  // 1024-pixel chunks per workgroup of k_m1_transform_generic
template <int PXSZ>
__global__ __launch_bounds__(256) void k_m1_transform_generic(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                              const TileDesc *__restrict__ tiles, TileSel sel,
                                                              uint32_t blocks_per_tile, const uint32_t *__restrict__ sums,
                                                              uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t *__restrict__ nlh, uint32_t nlh_slots) {
    bw_prio();
    __shared__ uint32_t s_red[4][6];
    NlAcc hacc = nlacc_zero();
    const uint32_t tile = vtile(sel, blockIdx.x / blocks_per_tile), chunk = blockIdx.x % blocks_per_tile;
    const TileDesc t = tiles[tile];
    const uint8_t *__restrict__ raster = rasters[t.img];
    // a workgroup covers TG_REPS * 1024 consecutive pixels of its tile: the tile descriptor and the predictor flags are
    // fetched once per thread, not once per four pixels (with one group per thread the kernel was a chain of three dependent
    // memory round trips per short-lived wave: latency-bound at a third of either its VALU or its HBM time)
    const int pr = pr_from_sums(sums + (uint64_t)tile * 4, PXSZ, t.w, t.h);
    for (uint32_t rep = 0; rep < TG_REPS; rep++) {
    const uint32_t i0 = ((chunk * TG_REPS + rep) * 256 + threadIdx.x) * 4;
    if (i0 >= t.n) break;
    const int useGrad = (pr >> 1) & 1, useG = pr & 1;
    uint32_t y = i0 / t.w, x = i0 - y * t.w;
    uint32_t onl = 0, orr = 0, og = 0, ob = 0, oa = 0;
    if (PXSZ == 3 && y > 0 && x > 0 && x + 4 < t.w) {
        // RGB interior group (the common case): the four pixels, their left neighbour and the five above them are 15 + 15
        // consecutive bytes: two unaligned 16-byte loads instead of 48 byte loads (the 16th byte belongs to pixel x + 4 of
        // the same tile row, so the loads stay inside the raster)
        typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
        const uint8_t *p = raster + (uint64_t)(t.y + y) * bpr + (uint64_t)(t.x + x) * 3 - 3;
        const u32x4_a1 c = *reinterpret_cast<const u32x4_a1 *>(p), u = *reinterpret_cast<const u32x4_a1 *>(p - bpr);
        const uint32_t cp[5] = {c.x & 0xFFFFFFu, __builtin_amdgcn_alignbyte(c.y, c.x, 3) & 0xFFFFFFu, __builtin_amdgcn_alignbyte(c.z, c.y, 2) & 0xFFFFFFu,
                                __builtin_amdgcn_alignbyte(c.w, c.z, 1) & 0xFFFFFFu, c.w & 0xFFFFFFu};
        const uint32_t up[5] = {u.x & 0xFFFFFFu, __builtin_amdgcn_alignbyte(u.y, u.x, 3) & 0xFFFFFFu, __builtin_amdgcn_alignbyte(u.z, u.y, 2) & 0xFFFFFFu,
                                __builtin_amdgcn_alignbyte(u.w, u.z, 1) & 0xFFFFFFu, u.w & 0xFFFFFFu};
        { const uint32_t pr_ = (pr & 3) & 3u; rgb_group_interior((int)(pr_ >> 1), (int)(pr_ & 1u), cp, up, onl, orr, og, ob); }  // (predictor flags are wave-uniform runtime values: one copy of the code, not four - the I-cache is shared by every kernel in flight)
    } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t nl = NL_NONE, zr = 0, zg = 0, zb = 0, za = 0;
        const uint32_t i = i0 + k;
        if (i < t.n && i > 0) {
            const uint8_t *p = raster + (uint64_t)(t.y + y) * bpr + (uint64_t)(t.x + x) * PXSZ;
            const uint32_t cur = load_px<PXSZ>(p);
            const bool row0 = y == 0, col0 = x == 0;
            const uint32_t L = col0 ? 0u : load_px<PXSZ>(p - PXSZ);
            const uint32_t U = row0 ? 0u : load_px<PXSZ>(p - bpr);
            const uint32_t UL = (row0 || col0) ? 0u : load_px<PXSZ>(p - bpr - PXSZ);
            m1_pixel<PXSZ>(cur, L, U, UL, row0, col0, useGrad, useG, nl, zr, zg, zb, za);
        }
        onl |= nl << (8 * k); orr |= zr << (8 * k); og |= zg << (8 * k); ob |= zb << (8 * k); oa |= za << (8 * k);
        if (++x == t.w) { x = 0; y++; }
    }
    }
    const uint64_t o = t.pbase + i0;
    *reinterpret_cast<uint32_t *>(planes + 0 * plane_stride + o) = onl;
    *reinterpret_cast<uint32_t *>(planes + 1 * plane_stride + o) = orr;
    *reinterpret_cast<uint32_t *>(planes + 2 * plane_stride + o) = og;
    *reinterpret_cast<uint32_t *>(planes + 3 * plane_stride + o) = ob;
    if (PXSZ == 4) *reinterpret_cast<uint32_t *>(planes + 4 * plane_stride + o) = oa;
    nlacc_add(hacc, onl, i0);
    }
    if (nlh) nlacc_commit<256>(hacc, s_red, nlh + ((uint64_t)tile * nlh_slots + chunk) * NLH_STRIDE);  // (nlh is a kernel argument: the whole workgroup takes the same side)
}

#ifndef XPNG_TR_ROWS
#define XPNG_TR_ROWS 8
#endif
constexpr uint32_t TR_ROWS = XPNG_TR_ROWS, TR_MAXW = 672, TR_PITCH = TR_MAXW * 4 + 32;  // bytes per LDS row (16-byte multiple)
static_assert((TR_ROWS * TR_MAXW / 4 + 255) / 256 <= 7 && TG_REPS <= 7, "NlAcc: a thread's groups per workgroup must fit its 3-bit fields");
__host__ __device__ inline uint32_t nlh_records(uint32_t w, uint32_t h, bool generic) {
    return generic ? (w * h + 1024u * TG_REPS - 1) / (1024u * TG_REPS) : (h + TR_ROWS - 1) / TR_ROWS;
}
// phase 2 of k_m1_transform_rgba, specialised on the tile's predictor flags so that no per-pixel branch on them remains
// one group of 4 consecutive pixels of a strip staged in LDS (RGBA): the five packed symbol dwords.  g = group index inside the
// strip, (yy, x0) = row inside the strip and column of its first pixel.
__device__ __forceinline__ void rgba_group(const int useGrad, const int useG, const uint8_t *rows, const TileDesc &t, uint64_t bpr, uint32_t y0, uint32_t first, uint32_t strip_px,
                                           uint32_t g, uint32_t yy, uint32_t x0, uint32_t &onl, uint32_t &orr, uint32_t &og, uint32_t &ob, uint32_t &oa) {
        const uint32_t j0 = g * 4;  // pixel index inside the strip
        uint32_t x = x0;
        onl = 0; orr = 0; og = 0; ob = 0; oa = 0;
        const uint32_t lr = yy + (y0 - first);  // LDS row of pixel row yy
        const uint32_t sh = (uint32_t)(((uint64_t)(t.y + first + lr) * bpr + (uint64_t)t.x * 4) & 15);
        const uint32_t *rowc = reinterpret_cast<const uint32_t *>(rows + lr * TR_PITCH + sh);
        const bool whole = x + 3 < t.w;                       // the four pixels share a row
        const bool row0 = (y0 + yy) == 0;                     // ... and it is the tile's first row (left-only prediction)
        if (j0 + 3 < strip_px && !row0) {
            // Rows below the first: byte-parallel arithmetic, then a 4x4 byte transpose into the planes.  Every such group goes
            // this way, including the one at the start of a row (column 0 predicts from above) and, for tile widths that are not
            // a multiple of 4, the one that straddles two rows: nearly every wave holds one of those, and the scalar path would
            // cost the whole wave its ~170 instructions.
            uint32_t c[5], u[5];
            bool col0[4] = {x == 0, false, false, false};
            if (whole && (bpr & 15) == 0) {
                const uint32_t *rowu = reinterpret_cast<const uint32_t *>(rows + (lr - 1) * TR_PITCH + sh);
                c[0] = x ? rowc[x - 1] : 0u;
                u[0] = x ? rowu[x - 1] : 0u;
                if (sh == 0) {  // 16-byte aligned group: one ds_read_b128 per row (dword reads at a 16-byte lane stride are 4-way bank conflicts)
                    const uint4 cc = *reinterpret_cast<const uint4 *>(rowc + x);
                    const uint4 uu = *reinterpret_cast<const uint4 *>(rowu + x);
                    c[1] = cc.x; c[2] = cc.y; c[3] = cc.z; c[4] = cc.w;
                    u[1] = uu.x; u[2] = uu.y; u[3] = uu.z; u[4] = uu.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) { c[k + 1] = rowc[x + k]; u[k + 1] = rowu[x + k]; }
                }
                uint32_t z[4], n4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) z[k] = m1_pixel_interior(useGrad, useG, c[k + 1], c[k], u[k + 1], u[k], n4[k], col0[k]);
                onl = n4[0] | (n4[1] << 8) | (n4[2] << 16) | (n4[3] << 24);
                const uint32_t t01lo = __builtin_amdgcn_perm(z[1], z[0], 0x05010400u);  // z0.b0 z1.b0 z0.b1 z1.b1
                const uint32_t t01hi = __builtin_amdgcn_perm(z[1], z[0], 0x07030602u);  // z0.b2 z1.b2 z0.b3 z1.b3
                const uint32_t t23lo = __builtin_amdgcn_perm(z[3], z[2], 0x05010400u);
                const uint32_t t23hi = __builtin_amdgcn_perm(z[3], z[2], 0x07030602u);
                orr = __builtin_amdgcn_perm(t23lo, t01lo, 0x05040100u);  // b0 of z0..z3
                og = __builtin_amdgcn_perm(t23lo, t01lo, 0x07060302u);   // b1
                ob = __builtin_amdgcn_perm(t23hi, t01hi, 0x05040100u);   // b2
                oa = __builtin_amdgcn_perm(t23hi, t01hi, 0x07060302u);   // b3
            } else {
                // gather pixel by pixel: rows may differ inside the group and (image width not a multiple of 4) so may each
                // row's 16-byte phase
                const uint32_t ph0 = (uint32_t)(((uint64_t)(t.y + first) * bpr + (uint64_t)t.x * 4) & 15), bl = (uint32_t)(bpr & 15);
                uint32_t z[4], n4[4], xx = x, ll = lr;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t *rc = reinterpret_cast<const uint32_t *>(rows + ll * TR_PITCH + ((ph0 + ll * bl) & 15u));
                    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rows + (ll - 1) * TR_PITCH + ((ph0 + (ll - 1) * bl) & 15u));
                    const uint32_t cur = rc[xx], L = xx ? rc[xx - 1] : 0u, U = ru[xx], UL = xx ? ru[xx - 1] : 0u;
                    z[k] = m1_pixel_interior(useGrad, useG, cur, L, U, UL, n4[k], xx == 0);
                    if (++xx == t.w) { xx = 0; ll++; }
                }
                onl = n4[0] | (n4[1] << 8) | (n4[2] << 16) | (n4[3] << 24);
                const uint32_t t01lo = __builtin_amdgcn_perm(z[1], z[0], 0x05010400u);
                const uint32_t t01hi = __builtin_amdgcn_perm(z[1], z[0], 0x07030602u);
                const uint32_t t23lo = __builtin_amdgcn_perm(z[3], z[2], 0x05010400u);
                const uint32_t t23hi = __builtin_amdgcn_perm(z[3], z[2], 0x07030602u);
                orr = __builtin_amdgcn_perm(t23lo, t01lo, 0x05040100u);
                og = __builtin_amdgcn_perm(t23lo, t01lo, 0x07060302u);
                ob = __builtin_amdgcn_perm(t23hi, t01hi, 0x05040100u);
                oa = __builtin_amdgcn_perm(t23hi, t01hi, 0x07060302u);
            }
        } else {  // the tile's first row and the last, partial group of a strip: pixel by pixel, every case
            uint32_t y = yy;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t nl = NL_NONE, zr = 0, zg = 0, zb = 0, za = 0;
                if (j0 + k < strip_px && (y0 + y + x) != 0) {
                    const uint32_t l2 = y + (y0 - first);
                    const uint32_t s2 = (uint32_t)(((uint64_t)(t.y + first + l2) * bpr + (uint64_t)t.x * 4) & 15);
                    const uint32_t s1 = l2 ? (uint32_t)(((uint64_t)(t.y + first + l2 - 1) * bpr + (uint64_t)t.x * 4) & 15) : 0u;
                    const uint32_t *rc = reinterpret_cast<const uint32_t *>(rows + l2 * TR_PITCH + s2);
                    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rows + (l2 ? l2 - 1 : 0) * TR_PITCH + s1);
                    const bool r0 = (y0 + y) == 0, c0 = x == 0;
                    const uint32_t cur = rc[x], L = c0 ? 0u : rc[x - 1], U = r0 ? 0u : ru[x], UL = (r0 || c0) ? 0u : ru[x - 1];
                    m1_pixel<4>(cur, L, U, UL, r0, c0, useGrad, useG, nl, zr, zg, zb, za);
                }
                onl |= nl << (8 * k); orr |= zr << (8 * k); og |= zg << (8 * k); ob |= zb << (8 * k); oa |= za << (8 * k);
                if (++x == t.w) { x = 0; y++; }
            }
        }
}

__device__ __forceinline__ void transform_phase2(const int useGrad, const int useG, const uint8_t *rows, const TileDesc &t, uint64_t bpr, uint32_t y0, uint32_t first,
                                                 uint32_t nrows, uint8_t *__restrict__ planes, uint64_t plane_stride, NlAcc &hacc) {
    const uint32_t strip_px = nrows * t.w, groups = (strip_px + 3) >> 2;
    // (row, column) of a thread's first group by one division, then advanced incrementally: +1024 pixels per iteration
    uint32_t yy = (threadIdx.x * 4) / t.w, x0 = threadIdx.x * 4 - yy * t.w;
    const uint32_t dy = 1024 / t.w, dx = 1024 - dy * t.w;
    for (uint32_t g = threadIdx.x; g < groups; g += 256, yy += dy, x0 += dx) {
        if (x0 >= t.w) { x0 -= t.w; yy++; }
        uint32_t onl, orr, og, ob, oa;
        rgba_group(useGrad, useG, rows, t, bpr, y0, first, strip_px, g, yy, x0, onl, orr, og, ob, oa);
        const uint32_t j0 = g * 4;
        const uint64_t o = t.pbase + (uint64_t)y0 * t.w + j0;
        *reinterpret_cast<uint32_t *>(planes + 0 * plane_stride + o) = onl;
        *reinterpret_cast<uint32_t *>(planes + 1 * plane_stride + o) = orr;
        *reinterpret_cast<uint32_t *>(planes + 2 * plane_stride + o) = og;
        *reinterpret_cast<uint32_t *>(planes + 3 * plane_stride + o) = ob;
        *reinterpret_cast<uint32_t *>(planes + 4 * plane_stride + o) = oa;
        nlacc_add(hacc, onl, y0 * t.w + j0);
    }
}

// --------------------------------------------------------------------------------------------------
// K2 (fast form, RGBA)  LDS-staged per-pixel transform.  One 256-thread workgroup = a strip of TR_ROWS rows of one tile:
//   phase 1  rows y0-1 .. y0+R-1 of the tile are copied global -> LDS with coalesced 16-byte loads (the halo row is the
//            only re-read: 1/TR_ROWS of the traffic); every row keeps its global 16-byte phase, so a pixel is one aligned
//            LDS dword;
//   phase 2  one thread = 4 consecutive pixels of the tile in raster order: cur / left / up / up-left come from LDS
//            (10 dword reads per 4 pixels when the group sits in one row), the five symbol bytes of the 4 pixels are
//            packed into one dword per plane and stored coalesced (256 B per wave instruction).
// Tiles wider than TR_MAXW pixels (images narrower / flatter than 444 px) use the generic kernel.

__global__ __launch_bounds__(256) void k_m1_transform_rgba(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                           uint64_t raster_bytes, const TileDesc *__restrict__ tiles, TileSel sel,
                                                           uint32_t strips_per_tile, const uint32_t *__restrict__ sums,
                                                           uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t nblocks,
                                                           uint32_t *__restrict__ nlh, uint32_t nlh_slots) {
    bw_prio();
    __shared__ __align__(16) uint8_t rows[(TR_ROWS + 1) * TR_PITCH];
    __shared__ uint32_t s_red[4][6];
    // Consecutive block ids round-robin over the 8 XCDs, each with its own L2; consecutive strips of a tile share a halo row.
    // The grid is padded to a multiple of 8 and re-read so that each XCD works through a contiguous run of strips.
    const uint32_t bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (bid >= nblocks) return;
    const uint32_t tile = vtile(sel, bid / strips_per_tile), strip = bid % strips_per_tile;
    const TileDesc t = tiles[tile];
    const uint32_t y0 = strip * TR_ROWS;
    if (y0 >= t.h) return;
    const uint8_t *__restrict__ raster = rasters[t.img];
    const uint32_t nrows = min(TR_ROWS, t.h - y0);
    const uint32_t first = y0 ? y0 - 1 : 0, lrows = y0 + nrows - first;  // rows staged: [first, first + lrows)
    // ---- phase 1: global -> LDS, 16 bytes per lane per load
    {
        const uint64_t g0 = (uint64_t)(t.y + first) * bpr + (uint64_t)t.x * 4;  // first staged byte
        const uint32_t row_bytes = t.w * 4;
        const uint32_t chunks = (uint32_t)(((g0 & 15) + row_bytes + 15) >> 4) + ((bpr & 15) ? 1 : 0);  // per row, upper bound (<= 170)
        // All of a thread's loads are issued before the first LDS store: ~5 x 16 B in flight per thread (Little's law:
        // 6 TB/s x ~2 us of loaded latency needs ~50 KB in flight per CU).
        // [r4] Which (row, chunk) a load covers is fixed by the thread's index - the workgroup's lower half takes the even staged
        // rows, the upper half the odd ones, thread t & 127 chunk t & 127 (+ 128 for the widest tiles) - so no load needs a
        // division: rounds 1-3 cut a flat chunk index into (row, chunk) with a float reciprocal and three 32-bit multiplies (a
        // quarter of the full vector rate each) per load, a third of the kernel's vector instructions.
        constexpr int KR = (TR_ROWS + 2) / 2, KC = (TR_MAXW * 4 / 16 + 2 + 127) / 128;  // rows per half, chunk passes
        static_assert(KC == 2, "chunks of a row fit two passes of 128 threads");
        const uint32_t half = threadIdx.x >> 7, c0 = threadIdx.x & 127u;
        uint4 v[KR][KC];
        bool ok[KR][KC];
        uint64_t gr = g0 + (uint64_t)half * bpr;  // first byte of this thread's first row segment
#pragma unroll
        for (int kr = 0; kr < KR; kr++, gr += 2 * bpr) {
            const uint32_t r = half + 2u * kr;
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                const uint32_t ch = c0 + 128u * kc;
                const uint64_t a = (gr & ~15ull) + (uint64_t)ch * 16;  // 16-byte aligned chunk
                ok[kr][kc] = r < lrows && ch < chunks && a < gr + row_bytes;
                if (ok[kr][kc]) {
                    if (a + 16 <= raster_bytes) v[kr][kc] = *reinterpret_cast<const uint4 *>(raster + a);
                    else {  // last chunk of the raster: stay inside the allocation
                        uint32_t w4[4] = {0, 0, 0, 0};
                        for (uint32_t q = 0; q < 4; q++) if (a + 4 * q + 4 <= raster_bytes) w4[q] = *reinterpret_cast<const uint32_t *>(raster + a + 4 * q);
                        v[kr][kc] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int kr = 0; kr < KR; kr++)
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
                if (ok[kr][kc]) *reinterpret_cast<uint4 *>(rows + (half + 2u * kr) * TR_PITCH + (c0 + 128u * kc) * 16) = v[kr][kc];
    }
    __syncthreads();
    // ---- phase 2 (dispatch once per workgroup on the tile's predictor flags)
    { const uint32_t pr_ = (pr_from_sums(sums + (uint64_t)tile * 4, 4, t.w, t.h) & 3) & 3u; NlAcc hacc = nlacc_zero(); transform_phase2((int)(pr_ >> 1), (int)(pr_ & 1u), rows, t, bpr, y0, first, nrows, planes, plane_stride, hacc);
      XPNG_BURN(g_burn_tr, threadIdx.x + pr_, reinterpret_cast<uint32_t *>(planes));
      if (nlh) nlacc_commit<256>(hacc, s_red, nlh + ((uint64_t)tile * nlh_slots + strip) * NLH_STRIDE); }
}

// --------------------------------------------------------------------------------------------------
// K2 (fast form, RGB)  the same LDS-staged transform for 3-byte pixels (14 of the reference's 17 corpus images, and the only
// input of level 2).  Phase 1 is the RGBA kernel's: rows y0-1 .. y0+R-1 of the tile leave HBM as coalesced, 16-byte aligned
// loads and keep their global 16-byte phase in LDS.  Phase 2: a thread owns 4 consecutive pixels of the tile; the 15 bytes
// "left neighbour + 4 pixels" of its row and of the row above are two 16-byte LDS reads at byte granularity (lane stride 12
// bytes: conflict-free), the pixels are cut out with v_alignbyte and go through the byte-parallel arithmetic with the alpha
// byte forced to 255 on both sides (it predicts itself: zero residual).  Four planes, one aligned dword store each.
constexpr uint32_t TR3_LDS_PAD = 16;  // the read of a column-0 group starts 3 bytes in front of its row
// the same for RGB (four packed symbol dwords); g0b = global byte offset of the strip's first staged byte (its 16-byte phase)
__device__ __forceinline__ void rgb_group(const int useGrad, const int useG, const uint8_t *rows, const TileDesc &t, uint64_t bpr, uint64_t g0, uint32_t y0, uint32_t first, uint32_t strip_px,
                                          uint32_t g, uint32_t yy, uint32_t x0, uint32_t &onl, uint32_t &orr, uint32_t &og, uint32_t &ob) {
    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t ph0 = (uint32_t)(g0 & 15), bl = (uint32_t)(bpr & 15);
    auto row_off = [&](uint32_t lr) { return TR3_LDS_PAD + lr * TR_PITCH + ((ph0 + lr * bl) & 15u); };  // LDS byte offset of column 0 of staged row lr
    auto px_at = [&](uint32_t lr, uint32_t x) {  // one pixel out of LDS (two aligned dwords)
        const uint32_t a = row_off(lr) + 3 * x;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(rows + (a & ~3u));
        return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u) & 0xFFFFFFu;
    };
        const uint32_t j0 = g * 4;
        uint32_t x = x0;
        onl = 0; orr = 0; og = 0; ob = 0;
        const uint32_t lr = yy + (y0 - first);
        const bool row0 = (y0 + yy) == 0;
        if (j0 + 3 < strip_px && !row0) {
            uint32_t z[4], n4[4];
            if (x + 3 < t.w) {  // the four pixels share a row: 15 + 15 bytes
                const uint32_t a = row_off(lr) + 3 * x - 3;
                const u32x4_a1 c = *reinterpret_cast<const u32x4_a1 *>(rows + a), u = *reinterpret_cast<const u32x4_a1 *>(rows + a - TR_PITCH - ((ph0 + lr * bl) & 15u) + ((ph0 + (lr - 1) * bl) & 15u));
                const uint32_t cp[5] = {c.x & 0xFFFFFFu, __builtin_amdgcn_alignbyte(c.y, c.x, 3) & 0xFFFFFFu, __builtin_amdgcn_alignbyte(c.z, c.y, 2) & 0xFFFFFFu,
                                        __builtin_amdgcn_alignbyte(c.w, c.z, 1) & 0xFFFFFFu, c.w & 0xFFFFFFu};
                const uint32_t up[5] = {u.x & 0xFFFFFFu, __builtin_amdgcn_alignbyte(u.y, u.x, 3) & 0xFFFFFFu, __builtin_amdgcn_alignbyte(u.z, u.y, 2) & 0xFFFFFFu,
                                        __builtin_amdgcn_alignbyte(u.w, u.z, 1) & 0xFFFFFFu, u.w & 0xFFFFFFu};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    z[k] = m1_pixel_interior(useGrad, useG, cp[k + 1] | 0xFF000000u, cp[k] | 0xFF000000u, up[k + 1] | 0xFF000000u, up[k] | 0xFF000000u, n4[k], k == 0 && x == 0);
            } else {            // a group that straddles two rows (tile width not a multiple of 4): pixel by pixel
                uint32_t xx = x, ll = lr;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t cur = px_at(ll, xx), L = xx ? px_at(ll, xx - 1) : 0u, U = px_at(ll - 1, xx), UL = xx ? px_at(ll - 1, xx - 1) : 0u;
                    z[k] = m1_pixel_interior(useGrad, useG, cur | 0xFF000000u, L | 0xFF000000u, U | 0xFF000000u, UL | 0xFF000000u, n4[k], xx == 0);
                    if (++xx == t.w) { xx = 0; ll++; }
                }
            }
            onl = n4[0] | (n4[1] << 8) | (n4[2] << 16) | (n4[3] << 24);
            const uint32_t t01lo = __builtin_amdgcn_perm(z[1], z[0], 0x05010400u), t01hi = __builtin_amdgcn_perm(z[1], z[0], 0x07030602u);
            const uint32_t t23lo = __builtin_amdgcn_perm(z[3], z[2], 0x05010400u), t23hi = __builtin_amdgcn_perm(z[3], z[2], 0x07030602u);
            orr = __builtin_amdgcn_perm(t23lo, t01lo, 0x05040100u);
            og = __builtin_amdgcn_perm(t23lo, t01lo, 0x07060302u);
            ob = __builtin_amdgcn_perm(t23hi, t01hi, 0x05040100u);
        } else {  // the tile's first row and the last, partial group of a strip: pixel by pixel, every case
            uint32_t y = yy;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t nl = NL_NONE, zr = 0, zg = 0, zb = 0, za = 0;
                if (j0 + k < strip_px && (y0 + y + x) != 0) {
                    const uint32_t l2 = y + (y0 - first);
                    const bool r0 = (y0 + y) == 0, c0 = x == 0;
                    const uint32_t cur = px_at(l2, x), L = c0 ? 0u : px_at(l2, x - 1), U = r0 ? 0u : px_at(l2 - 1, x), UL = (r0 || c0) ? 0u : px_at(l2 - 1, x - 1);
                    m1_pixel<3>(cur, L, U, UL, r0, c0, useGrad, useG, nl, zr, zg, zb, za);
                }
                onl |= nl << (8 * k); orr |= zr << (8 * k); og |= zg << (8 * k); ob |= zb << (8 * k);
                if (++x == t.w) { x = 0; y++; }
            }
        }
}

__device__ __forceinline__ void transform_phase2_rgb(const int useGrad, const int useG, const uint8_t *rows, const TileDesc &t, uint64_t bpr, uint64_t g0, uint32_t y0, uint32_t first,
                                                     uint32_t nrows, uint8_t *__restrict__ planes, uint64_t plane_stride, NlAcc &hacc) {
    const uint32_t strip_px = nrows * t.w, groups = (strip_px + 3) >> 2;
    uint32_t yy = (threadIdx.x * 4) / t.w, x0 = threadIdx.x * 4 - yy * t.w;
    const uint32_t dy = 1024 / t.w, dx = 1024 - dy * t.w;
    for (uint32_t g = threadIdx.x; g < groups; g += 256, yy += dy, x0 += dx) {
        if (x0 >= t.w) { x0 -= t.w; yy++; }
        uint32_t onl, orr, og, ob;
        rgb_group(useGrad, useG, rows, t, bpr, g0, y0, first, strip_px, g, yy, x0, onl, orr, og, ob);
        const uint32_t j0 = g * 4;
        const uint64_t o = t.pbase + (uint64_t)y0 * t.w + j0;
        *reinterpret_cast<uint32_t *>(planes + 0 * plane_stride + o) = onl;
        *reinterpret_cast<uint32_t *>(planes + 1 * plane_stride + o) = orr;
        *reinterpret_cast<uint32_t *>(planes + 2 * plane_stride + o) = og;
        *reinterpret_cast<uint32_t *>(planes + 3 * plane_stride + o) = ob;
        nlacc_add(hacc, onl, y0 * t.w + j0);
    }
}

__global__ __launch_bounds__(256) void k_m1_transform_rgb(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                          uint64_t raster_bytes, const TileDesc *__restrict__ tiles, TileSel sel,
                                                          uint32_t strips_per_tile, const uint32_t *__restrict__ sums,
                                                          uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t nblocks,
                                                          uint32_t *__restrict__ nlh, uint32_t nlh_slots) {
    bw_prio();
    __shared__ __align__(16) uint8_t rows[TR3_LDS_PAD + (TR_ROWS + 1) * TR_PITCH + 16];
    __shared__ uint32_t s_red[4][6];
    const uint32_t bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware: consecutive strips of a tile on one XCD
    if (bid >= nblocks) return;
    const uint32_t tile = vtile(sel, bid / strips_per_tile), strip = bid % strips_per_tile;
    const TileDesc t = tiles[tile];
    const uint32_t y0 = strip * TR_ROWS;
    if (y0 >= t.h) return;
    const uint8_t *__restrict__ raster = rasters[t.img];
    const uint32_t nrows = min(TR_ROWS, t.h - y0);
    const uint32_t first = y0 ? y0 - 1 : 0, lrows = y0 + nrows - first;
    const uint64_t g0 = (uint64_t)(t.y + first) * bpr + (uint64_t)t.x * 3;  // first staged byte
    {   // ---- phase 1: global -> LDS, 16 bytes per lane per load, every load in flight before the first LDS store
        const uint32_t row_bytes = t.w * 3;
        const uint32_t chunks = ((15u + row_bytes + 15u) >> 4) + 1;  // per row, upper bound for any phase
        constexpr int LD = ((TR_ROWS + 1) * (TR_MAXW * 3 / 16 + 3) + 255) / 256;
        const uint32_t total_chunks = lrows * chunks;
        const float inv_chunks = 1.0f / (float)chunks;
        uint4 v[LD];
        uint32_t dst[LD];
#pragma unroll
        for (int k = 0; k < LD; k++) {
            const uint32_t idx = threadIdx.x + 256u * k;
            dst[k] = ~0u;
            if (idx < total_chunks) {
                uint32_t r = (uint32_t)((float)idx * inv_chunks);
                if (r * chunks > idx) r--;
                if ((r + 1) * chunks <= idx) r++;
                const uint32_t ch = idx - r * chunks;
                const uint64_t gr = g0 + (uint64_t)r * bpr;
                const uint64_t a = (gr & ~15ull) + (uint64_t)ch * 16;
                if (a < gr + row_bytes) {
                    dst[k] = TR3_LDS_PAD + r * TR_PITCH + ch * 16;
                    if (a + 16 <= raster_bytes) v[k] = *reinterpret_cast<const uint4 *>(raster + a);
                    else {  // last chunk of the raster: stay inside the allocation (a raster need not be a multiple of 4 bytes long)
                        uint32_t w4[4] = {0, 0, 0, 0};
                        for (uint32_t q = 0; q < 16; q++) if (a + q < raster_bytes) w4[q >> 2] |= (uint32_t)raster[a + q] << (8 * (q & 3));
                        v[k] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LD; k++) if (dst[k] != ~0u) *reinterpret_cast<uint4 *>(rows + dst[k]) = v[k];
    }
    __syncthreads();
    { const uint32_t pr_ = (pr_from_sums(sums + (uint64_t)tile * 4, 3, t.w, t.h) & 3) & 3u; NlAcc hacc = nlacc_zero(); transform_phase2_rgb((int)(pr_ >> 1), (int)(pr_ & 1u), rows, t, bpr, g0, y0, first, nrows, planes, plane_stride, hacc);
      if (nlh) nlacc_commit<256>(hacc, s_red, nlh + ((uint64_t)tile * nlh_slots + strip) * NLH_STRIDE); }
}

// --------------------------------------------------------------------------------------------------
// --------------------------------------------------------------------------------------------------
// K3  stream formation.  One 1024-thread workgroup walks one tile in raster order, 1024 pixels per step.
// Three serial couplings of the reference loop become wave-level prefix operations:
//   (i)  pl = nl of the previous CODED pixel            -> ballot + "highest set bit below me" + carry
//   (ii) append position inside context stream cx[pl]   -> one ballot/popcount per context (9)
//   (iii) bit cursor of k (3*nl bits per coded pixel)   -> wave inclusive scan + cross-wave offsets,
//        bits are OR-ed MSB-first into an LDS word window and spliced into k with a carried partial word.
// Outputs: ctx streams (their lengths and places are known beforehand: k_m1_lens), k words + count.   grid = tiles, block = 1024.
// THREADS = 1024 for a few tiles (shortest serial walk per tile); 256 for large batches: the chain kernels of other
// batches in flight leave few CUs with room for a 16-wave workgroup, but almost all have room for a 4-wave one.
#ifndef XPNG_ST_WAVES
#define XPNG_ST_WAVES 0   // waves per SIMD the routing kernel is compiled for (0: whatever its registers allow - 6)
#endif
template <int PXSZ, int ST_THREADS>
__global__ __launch_bounds__(ST_THREADS, (XPNG_ST_WAVES && ST_THREADS == 256) ? XPNG_ST_WAVES : 1) void k_m1_streams(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                           const TileDesc *__restrict__ tiles, TileSel sel,
                                                           const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                           uint8_t *__restrict__ scratch, const uint32_t *__restrict__ ctx_n,
                                                           uint32_t *__restrict__ k_n) {
    bw_prio();
    // One workgroup walks a tile in raster order, ST_THREADS * 4 pixels per iteration: a lane owns 4 consecutive pixels
    // (one dword of each plane).  Per iteration: the previous-coded-nl chain (pl), a packed prefix sum of the nine
    // per-context counts (three 10-bit fields per word, three DPP scans), one scan of the bit lengths, byte stores into the
    // context streams and up to 96 bits per lane OR-ed into an LDS bit window that is flushed to k as whole words.
    const uint32_t tile = vtile(sel, blockIdx.x);
    const TileDesc t = tiles[tile];
    const uint8_t *__restrict__ raster = rasters[t.img];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t *pnl = reinterpret_cast<const uint32_t *>(planes + t.pbase), *pr_ = reinterpret_cast<const uint32_t *>(planes + plane_stride + t.pbase);
    const uint32_t *pg = reinterpret_cast<const uint32_t *>(planes + 2 * plane_stride + t.pbase), *pb = reinterpret_cast<const uint32_t *>(planes + 3 * plane_stride + t.pbase);
    uint8_t *sc = scratch + t.sbase;
    uint32_t *kw = reinterpret_cast<uint32_t *>(sc + off_kw(t.n));

    constexpr int ST_WAVES = ST_THREADS / 64, ST_PX = ST_THREADS * 4, ST_WORDS = ST_PX * 24 / 32 + 4, ST_ROUNDS = (ST_WORDS + ST_THREADS - 1) / ST_THREADS;
    __shared__ uint32_t s_bits[2][ST_WORDS];  // two bit windows, used alternately: an iteration's window is cleared by the threads that flush it, its partial last word opens the other one [r4: no barrier between the flush and the next iteration]
    __shared__ uint32_t s_run_cnt[9];  // next free byte of every context stream, relative to the tile's scratch (starts at the stream's place)
    __shared__ uint32_t s_wave_cnt[ST_WAVES][9];
    __shared__ volatile uint32_t s_base[ST_WAVES][16];
    __shared__ uint32_t s_wave_bits[ST_WAVES];
    __shared__ uint32_t s_wave_last[ST_WAVES];

    // first pixel: 8*PXSZ raw bits at the head of k (libxpng.c:547)
    const uint8_t *p0 = raster + (uint64_t)t.y * bpr + (uint64_t)t.x * PXSZ;
    uint32_t run_bits, wbase, run_pl = 0;  // uniform across the workgroup
    for (int j = tid; j < 2 * ST_WORDS; j += ST_THREADS) (&s_bits[0][0])[j] = 0;
    if (tid < 9) s_run_cnt[tid] = (uint32_t)off_ctx(t.n, ctx_n + (uint64_t)tile * 9, (int)tid);
    __syncthreads();
    if (PXSZ == 4) {
        if (tid == 0) kw[0] = ((uint32_t)p0[0] << 24) | ((uint32_t)p0[1] << 16) | ((uint32_t)p0[2] << 8) | p0[3];
        run_bits = 32; wbase = 1;
    } else {
        if (tid == 0) s_bits[0][0] = ((uint32_t)p0[0] << 24) | ((uint32_t)p0[1] << 16) | ((uint32_t)p0[2] << 8);
        run_bits = 24; wbase = 0;
    }
    const uint64_t lt = lanemask_lt();
    // field of context c in the packed counters: word c / 3, bits [10 * (c % 3), +10)
    const uint32_t myq = __umul24(lane, 11u) >> 5, mysh = __umul24(lane - __umul24(myq, 3u), 10u);

    // planes are read one iteration ahead (dwords; the planes carry >= 192 bytes of slack behind a tile)
    uint32_t nx_nl = 0xFFFFFFFFu, nx_r = 0, nx_g = 0, nx_b = 0;
    if (4 * tid < t.n) { nx_nl = pnl[tid]; nx_r = pr_[tid]; nx_g = pg[tid]; nx_b = pb[tid]; }

    uint32_t wpar = 0;  // which bit window this iteration fills
    for (uint32_t i0 = 0; i0 < t.n; i0 += ST_PX, wpar ^= 1) {
        const uint32_t i = i0 + 4 * tid;
        uint32_t nl4 = nx_nl;
        const uint32_t r4 = nx_r, g4 = nx_g, b4 = nx_b;
        {
            const uint32_t in = i + ST_PX;
            nx_nl = 0xFFFFFFFFu;
            if (in < t.n) { nx_nl = pnl[in >> 2]; nx_r = pr_[in >> 2]; nx_g = pg[in >> 2]; nx_b = pb[in >> 2]; }
        }
        if (i < t.n && t.n - i < 4) nl4 |= 0xFFFFFFFFu << (8 * (t.n - i));  // pixels past the tile: not coded
        uint32_t nl[4], len[4];
        bool coded[4];
        uint32_t lastnl = NL_NONE, lane_len = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            nl[j] = (nl4 >> (8 * j)) & 255u;
            coded[j] = nl[j] != NL_NONE;
            lastnl = coded[j] ? nl[j] : lastnl;
            len[j] = coded[j] ? 3 * nl[j] : 0;
            lane_len += len[j];
        }
        const uint64_t mask = __ballot(nl4 != 0xFFFFFFFFu);
        const uint64_t lower = mask & lt;
        // (i) previous coded nl: inside the wave by a lane fetch, across waves through LDS
        const int src_last = mask ? 63 - __clzll((long long)mask) : 0;
        const uint32_t wave_last = __shfl(lastnl, src_last);
        if (lane == 0) s_wave_last[wv] = mask ? wave_last : NL_NONE;
        const int src_prev = lower ? 63 - __clzll((long long)lower) : 0;
        const uint32_t prev_in_wave = __shfl(lastnl, src_prev);
        // (iii) bit lengths: inclusive scan inside the wave
        const uint32_t incl = wave_scan_incl(lane_len);
        if (lane == 63) s_wave_bits[wv] = incl;
        __syncthreads();  // (A) wave_last / wave_bits visible

        uint32_t carry = run_pl, bit_base = 0, bits_total = 0, new_run_pl = run_pl;
        for (int w2 = 0; w2 < ST_WAVES; w2++) {
            const uint32_t wl = s_wave_last[w2], wb = s_wave_bits[w2];
            if (w2 < (int)wv) { if (wl != NL_NONE) carry = wl; bit_base += wb; }
            if (wl != NL_NONE) new_run_pl = wl;
            bits_total += wb;
        }
        // (ii) pl of each pixel and the lane's packed per-context counts
        uint32_t pl[4], sh[4], inc[4][3], w0 = 0, w1 = 0, w2s = 0;
        {
            uint32_t p = lower ? prev_in_wave : carry;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                pl[j] = p;
                const uint32_t q = __umul24(p, 11u) >> 5;  // (24-bit multiplies: full rate; a 32-bit v_mul_lo_u32 issues at a quarter of it)
                sh[j] = __umul24(p - __umul24(q, 3u), 10u);
                const uint32_t one = coded[j] ? 1u << sh[j] : 0u;
                inc[j][0] = q == 0 ? one : 0u; inc[j][1] = q == 1 ? one : 0u; inc[j][2] = q == 2 ? one : 0u;
                w0 += inc[j][0]; w1 += inc[j][1]; w2s += inc[j][2];
                p = coded[j] ? nl[j] : p;
            }
        }
        const uint32_t x0 = wave_scan_incl(w0), x1 = wave_scan_incl(w1), x2 = wave_scan_incl(w2s);
        {
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, 63),
                           t2 = (uint32_t)__builtin_amdgcn_readlane((int)x2, 63);
            const uint32_t word = myq == 0 ? t0 : (myq == 1 ? t1 : t2);
            if (lane < 9) s_wave_cnt[wv][lane] = (word >> mysh) & 1023u;
        }
        __syncthreads();  // (B) per-wave context counts visible
        if (lane < 9) {   // this wave's base position in each context stream (wave-private LDS row: DS ops of a wave stay in order)
            uint32_t base = s_run_cnt[lane];
            for (uint32_t w = 0; w < wv; w++) base += s_wave_cnt[w][lane];
            s_base[wv][lane] = base;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t e0 = x0 - w0, e1 = x1 - w1, e2 = x2 - w2s;  // exclusive lane prefixes, advanced pixel by pixel below
        uint32_t pos = (run_bits & 31) + bit_base + (incl - lane_len);
        const uint32_t wi = pos >> 5;
        pos &= 31;
        uint32_t fv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (coded[j]) {
                const uint32_t e = inc[j][0] ? e0 : (inc[j][1] ? e1 : e2);
                const uint32_t rank = (e >> sh[j]) & 1023u;
                sc[s_base[wv][pl[j]] + rank] = (uint8_t)nl[j];
            }
            e0 += inc[j][0]; e1 += inc[j][1]; e2 += inc[j][2];
            const uint32_t n1 = nl[j] & 15u;  // (NL_NONE has len 0: v is masked out below)
            fv[j] = len[j] ? ((((r4 >> (8 * j)) & 255u) << (2 * n1)) | (((g4 >> (8 * j)) & 255u) << n1) | ((b4 >> (8 * j)) & 255u)) : 0u;
        }
        {
            // The lane's bit string (its four fields, first pixel first, <= 96 bits) left-aligned in m0:m1:m2, built from the LAST field
            // backwards: every field is prepended by one funnel shift to the right per word (rounds 1-3 placed each field with a
            // 128-bit shift of its own: ~60 vector instructions per lane and iteration, a quarter of them at half rate), then the
            // string moves `pos` bits to the right into the four words it is OR-ed into.  (A field of length 0 is 0: the shift by
            // (32 - 0) & 31 = 0 of nothing is nothing.)
            uint32_t m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
            for (int j = 3; j >= 0; j--) {
                const uint32_t l = len[j];
                m2 = __builtin_amdgcn_alignbit(m1, m2, l);
                m1 = __builtin_amdgcn_alignbit(m0, m1, l);
                m0 = (m0 >> l) | (fv[j] << ((32u - l) & 31u));
            }
            const uint32_t f0 = m0 >> pos, f1 = __builtin_amdgcn_alignbit(m0, m1, pos), f2 = __builtin_amdgcn_alignbit(m1, m2, pos), f3 = __builtin_amdgcn_alignbit(m2, 0u, pos);
            uint32_t *win = s_bits[wpar];
            if (f0) atomicOr(&win[wi], f0);
            if (f1) atomicOr(&win[wi + 1], f1);
            if (f2) atomicOr(&win[wi + 2], f2);
            if (f3) atomicOr(&win[wi + 3], f3);
        }
        __syncthreads();  // (C) bit window complete

        const uint32_t nfull = ((run_bits & 31) + bits_total) >> 5;
        {
            // whole words leave for k; the thread that reads a word clears it; the one that holds the partial last word (index nfull)
            // moves it to word 0 of the OTHER window, which the flush of the iteration before this one left all zero
            uint32_t *win = s_bits[wpar], *nwin = s_bits[wpar ^ 1];
#pragma unroll
            for (int k = 0; k < ST_ROUNDS; k++) {
                const uint32_t j = tid + k * ST_THREADS;
                if (j < (uint32_t)ST_WORDS) {
                    const uint32_t v = win[j];
                    win[j] = 0u;
                    if (j < nfull) kw[wbase + j] = v;
                    else if (j == nfull) nwin[0] = v;
                }
            }
        }
        if (tid < 9) {
            uint32_t tot_c = 0;
            for (int w = 0; w < ST_WAVES; w++) tot_c += s_wave_cnt[w][tid];
            s_run_cnt[tid] += tot_c;
        }
        run_bits += bits_total; wbase += nfull; run_pl = new_run_pl;
        // (no barrier here: the next iteration ORs into the other window only behind its barriers (A) and (B), and the LDS words this
        //  iteration's tail reads - the per-wave counts, the window - are rewritten only behind barrier (A) of the next)
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t words = wbase;
        if (run_bits & 31) { kw[wbase] = s_bits[wpar][0]; words++; }  // BITSTREAM_END: tail is already left-aligned (it opened the window the next iteration would have filled)
        k_n[tile] = words;
    }
}

}  // namespace xpng
