// tile_container.hpp -- tile blob assembly + concatenation on the device.
//
// Reference: tail of enc_1_th (libxpng.c:556-568: k size word, blocks, raw-tile fallback, header word) and the
// in-order concatenation of xpng_store_T (libxpng.c:764-769).  Variable-length output => sizes, exclusive
// scan, gather.
#pragma once
#include "common.hpp"
#include "rans2.hpp"
#include "rans2_wide.hpp"

namespace xpng {

// K4  one wavefront per (tile, stream): the entropy stage of enc_1_th (libxpng.c:558-559).
// Stream c < 9: context stream c, alphabet 9, PROB_BITS 12.  Stream 9 (RGBA): alpha symbols = plane `a`
// from index 1, alphabet 256, PROB_BITS 15.      grid = work items * c_count, block = 64: streams [c_first, c_first + c_count)
// of the first grid / c_count work items of the enumeration.
// Two uses.  (1) A few tiles (one image): every stream of every tile, the whole entropy stage of the encode.  (2) Beside the wide
// form of a batch [r4]: ONE stream class - alpha, c_first = 9 - of the first work items of the size-sorted enumeration, i.e. of the
// biggest tiles.  A wide chain kernel lasts as long as its longest chain and a lane-per-state step takes ~400 cycles against ~176
// here: with the alpha streams of the biggest size class (>= 3/4 of the largest tile: 17 of a 4096^2 image's 81) a wavefront each,
// the wide launch of the rest ends with the 444 x 444 tiles' chains - two thirds of the longest.  The block is final when this
// kernel is done (header, states, table, raw fallback): prep[] (when given) says so and k_rans2_finish leaves it alone.
// (ONE kernel for both uses: with two callers the compiler stops inlining rans2_encode_block and its hand-scheduled loop becomes a
//  function call - the single-image encode went from 11.5 to 14.7 ms when a second kernel called it.)
__global__ __launch_bounds__(64) void k_rans2_encode(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t c_first, uint32_t c_count,
                                                     const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                     uint8_t *__restrict__ scratch, const uint32_t *__restrict__ ctx_n,
                                                     uint32_t *__restrict__ blk_sz, WPrep *__restrict__ prep, uint64_t *__restrict__ dbg) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cum[260];
    __shared__ EncSym tab[256];
    const uint32_t tile = vtile(sel, blockIdx.x / c_count), c = c_first + blockIdx.x % c_count;
    const TileDesc t = tiles[tile];
    uint8_t *sc = scratch + t.sbase;
    const uint8_t *in;
    uint32_t n, nominalN;
    int pb;
    if (c < 9) { in = sc + off_ctx(t.n, ctx_n + (uint64_t)tile * 9, (int)c); n = ctx_n[(uint64_t)tile * 9 + c]; nominalN = 9; pb = 12; }
    else { in = planes + 4 * plane_stride + t.pbase + 1; n = t.n - 1; nominalN = 256; pb = 15; }
    const uint32_t sz = rans2_encode_block(in, n, nominalN, pb, sc + off_blk(t.n, ctx_n + (uint64_t)tile * 9, (int)c), hist, cum, tab,
                                           dbg ? dbg + ((uint64_t)tile * 10 + c) * 8 : nullptr);
    if ((threadIdx.x & 63) == 0) { blk_sz[(uint64_t)tile * 10 + c] = sz; if (prep) prep[(uint64_t)tile * 10 + c].kind = 0; }
}

// K5a  per-tile size + header word.  One thread per tile.
__global__ void k_tile_sizes(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t total, int pxsz, uint32_t spt,
                             const uint32_t *__restrict__ sums, const uint32_t *__restrict__ k_n,
                             const uint32_t *__restrict__ blk_sz, uint32_t *__restrict__ tile_sz,
                             uint32_t *__restrict__ tile_hdr) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    uint64_t fsz = 4 + 4 + 4ull * k_n[tile];
    for (uint32_t c = 0; c < spt; c++) fsz += blk_sz[(uint64_t)tile * 10 + c];
    const uint64_t raw = (uint64_t)t.n * pxsz + 4;
    const int pr = pr_from_sums(sums + (uint64_t)tile * 4, pxsz, t.w, t.h);
    const uint32_t il = imglin(sel, tile);  // sizes are scanned per image, in tile order
    if (fsz < raw) { tile_sz[il] = (uint32_t)fsz; tile_hdr[il] = (1u << 28) + ((uint32_t)pr << 24) + (uint32_t)fsz; }  // libxpng.c:563-564
    else { tile_sz[il] = (uint32_t)raw; tile_hdr[il] = (uint32_t)raw; }                                                  // libxpng.c:566
}

// K5b  exclusive scan of tile sizes -> byte offsets inside each image's blob buffer.  One workgroup per image:
// off[img * (cnt + 1) + i], the image's total at index cnt and in totals[img].
__global__ __launch_bounds__(1024) void k_tile_offsets(const uint32_t *__restrict__ tile_sz_all, uint32_t cnt,
                                                       uint64_t *__restrict__ off_all, uint64_t *__restrict__ totals) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_base;
    const uint32_t *tile_sz = tile_sz_all + (uint64_t)blockIdx.x * cnt;
    uint64_t *off = off_all + (uint64_t)blockIdx.x * (cnt + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const uint32_t nt = blockDim.x;  // (256 in the launches: a 1024-thread workgroup needs 16 free wave slots on one CU at once and waited up to 5 ms for them beside the chain kernels of other pipeline slots - on the critical path between the last rANS block and the gather)
    for (uint32_t i0 = 0; i0 < cnt; i0 += nt) {
        const uint32_t i = i0 + tid;
        const uint64_t v = i < cnt ? tile_sz[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d);
            if ((int)lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        uint64_t base = s_base;
        for (uint32_t w2 = 0; w2 < wv; w2++) base += s_wave[w2];
        if (i < cnt) off[i] = base + incl - v;
        __syncthreads();
        if (tid == nt - 1) s_base = base + incl;
        __syncthreads();
    }
    if (tid == 0) { off[cnt] = s_base; totals[blockIdx.x] = s_base; }
}

__device__ __forceinline__ void block_copy(uint8_t *dst, const uint8_t *src, uint64_t bytes) {
    if ((((uintptr_t)dst | (uintptr_t)src) & 3) == 0) {
        const uint64_t words = bytes >> 2;
        for (uint64_t i = threadIdx.x; i < words; i += blockDim.x)
            reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        for (uint64_t i = (words << 2) + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
    } else {
        for (uint64_t i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
    }
}

// K5c  gather every tile's pieces to its final place.  grid = tiles, block = 256.
__global__ __launch_bounds__(256) void k_tile_gather(const uint8_t *const *__restrict__ rasters, uint64_t bpr, int pxsz,
                                                     const TileDesc *__restrict__ tiles, TileSel sel, uint32_t spt,
                                                     const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ k_n,
                                                     const uint32_t *__restrict__ ctx_n,
                                                     const uint32_t *__restrict__ blk_sz, const uint32_t *__restrict__ tile_hdr,
                                                     const uint64_t *__restrict__ off, uint8_t *const *__restrict__ blobs) {
    bw_prio();
    const uint32_t j = blockIdx.x, tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const uint8_t *__restrict__ raster = rasters[t.img];
    const uint32_t il = imglin(sel, tile);
    uint8_t *dst = blobs[t.img] + off[(uint64_t)il + il / sel.cnt];  // off holds cnt + 1 entries per image
    const uint32_t hdr = tile_hdr[il];
    if (threadIdx.x < 4) dst[threadIdx.x] = (uint8_t)(hdr >> (8 * threadIdx.x));
    if ((hdr >> 24) == 0) {  // raw tile: rows (libxpng.c:566-567)
        const uint64_t row = (uint64_t)t.w * pxsz;
        const uint8_t *src = raster + (uint64_t)t.y * bpr + (uint64_t)t.x * pxsz;
        for (uint32_t y = 0; y < t.h; y++) block_copy(dst + 4 + y * row, src + y * bpr, row);
        return;
    }
    const uint8_t *sc = scratch + t.sbase;
    const uint32_t ksz = 4 + 4 * k_n[tile];
    if (threadIdx.x < 4) dst[4 + threadIdx.x] = (uint8_t)(ksz >> (8 * threadIdx.x));
    block_copy(dst + 8, sc + off_kw(t.n), ksz - 4);
    uint64_t o = 4 + ksz;
    for (uint32_t c = 0; c < spt; c++) {
        const uint32_t sz = blk_sz[(uint64_t)tile * 10 + c];
        block_copy(dst + o, sc + off_blk(t.n, ctx_n + (uint64_t)tile * 9, (int)c), sz);
        o += sz;
    }
}

}  // namespace xpng
