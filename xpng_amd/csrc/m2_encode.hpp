// m2_encode.hpp -- mode-2 (XPNG_COMPRESSION_TYPE_SLOW, RGB only) tile ENCODE kernels for gfx950.
//
// Reference path restated: enc_2_th (libxpng.c:645-686) = is_single_color (628-643) + is_grayscale (583-626) + pp_rgbx +
// ENC/ENC4 (33-44) + 17x compress_block (rANS v1, 160-260) + tile container.  Stages:
//
//   k_m2_classify   per tile: "all pixels equal the first" / "every pixel has R == G == B" (two OR-reductions)
//   k_chooser + k_m1_transform_generic<3>   the same predictor chooser and per-pixel arithmetic as mode 1 (ENC == M1ENC for RGB)
//   k_m2_streams    routing: nl -> context stream cx[pl] (as mode 1) AND residuals -> magnitude-class stream st[nl]
//                   (nl=1: one 3-bit symbol, nl=2: one 6-bit symbol, nl>=3: three symbols), all by wave ballots
//   k_m2_gray_syms  gray tiles only: the four candidate predictor streams (p1x, p1y, p2a, p3a)
//   k_rans1_encode  one wave per (tile, stream): rANS v1 = backwards over the symbols, state1 spills before state0, words are
//                   emitted in encode order and reversed at gather time; freq table (or the raw symbols) go to the tile's
//                   shared bit stream `b` as a "piece"
//   k_m2_select     tile type, gray candidate choice (first minimum), sizes, raw fallbacks
//   k_m2_bits       wave-parallel splice of the pieces into `b` at arbitrary bit offsets
//   k_tile_offsets + k_m2_gather   per-image scan and final placement
#pragma once
#include "common.hpp"
#include "m1_encode.hpp"
#include "rans2.hpp"
#include "tile_container.hpp"

namespace xpng {

constexpr uint32_t M2_STREAMS = 17, M2_SLOTS = 21;  // 9 context + 8 class streams; +4 gray candidates
__host__ __device__ inline uint32_t m2_nominal(uint32_t s) {  // alphabet of stream s (libxpng.c:668-669)
    if (s < 9) return 9;
    if (s >= 17) return 256;
    const uint32_t v = s - 8;  // class 1..8
    return v < 3 ? (1u << (3 * v)) : (1u << v);
}
// per-tile scratch layout for mode 2 (bytes, relative to the tile's m2 scratch base):
//   [the symbol streams, back to back][block slots][table pieces][bit stream b]
// The stream lengths are known before a symbol is written (k_m2_count: a histogram of the nl plane, as for mode 1; a gray tile's
// four candidates hold n - 1 symbols each) - in a decode they are the block headers' symbol counts - so every stream gets the
// room its length needs: a colour tile's nine context streams share n - 1 symbols, its eight class streams at most 3 (n - 1),
// and the region is 4 n bytes + slack instead of 29 n (nine + two streams of n and six of 3 n in rounds 1-3).
// cnt: the tile's M2_SLOTS stream lengths (stream_n + tile * M2_SLOTS; absent streams 0).
__host__ __device__ inline uint64_t m2_slot(uint32_t m) { return rup((uint64_t)m + 32, 64); }
__host__ __device__ inline uint64_t m2_streams_region(uint32_t n) { return rup(4ull * n + 21 * 96, 256); }
__host__ __device__ inline uint64_t m2_off_stream(uint32_t, const uint32_t *cnt, uint32_t s) {
    uint64_t o = 0;
    for (uint32_t i = 0; i < s; i++) o += m2_slot(cnt[i]);
    return o;
}
// Block slots (the words a stream's rANS chain emits: <= 15 bits per symbol, + states and slack) are laid out back to back, each
// sized by ITS stream's symbol count cnt[k] (known once k_m2_streams / k_m2_gray_syms have run), not by the stream's capacity: a
// colour tile's 17 streams hold at most n context + 3 n class symbols together, a gray tile's four candidates n each, so the region
// is 8 n bytes + slack instead of the 58 n of one worst-case slot per stream (5.7 -> 2.6 GB of workspace per 4096^2 image).
__host__ __device__ inline uint64_t m2_blk_cap(uint32_t m) { return rup(2ull * m + 512, 256); }
__host__ __device__ inline uint64_t m2_blk_region(uint32_t n) { return rup(8ull * n + 21 * 768, 256); }
__host__ __device__ inline uint64_t m2_off_blk(uint32_t n, const uint32_t *cnt, uint32_t s) {  // cnt: the tile's M2_SLOTS stream lengths
    uint64_t o = m2_streams_region(n);
    if (s >= 17) return o + (uint64_t)(s - 17) * m2_blk_cap(n);  // gray candidates (a gray tile has no colour blocks)
    for (uint32_t k = 0; k < s; k++) o += m2_blk_cap(cnt[k]);
    return o;
}
__host__ __device__ inline uint64_t m2_off_piece(uint32_t n, uint32_t s) {  // table bits of stream s: <= 256 * 16 bits
    return m2_streams_region(n) + m2_blk_region(n) + (uint64_t)s * 640;
}
__host__ __device__ inline uint64_t m2_off_bits(uint32_t n) { return m2_off_piece(n, M2_SLOTS); }
__host__ __device__ inline uint64_t m2_bits_cap(uint32_t n) { return rup(3ull * n + 21 * 640 + 256, 256); }
__host__ __device__ inline uint64_t m2_tile_scratch(uint32_t n) { return m2_off_bits(n) + m2_bits_cap(n); }

struct M2Blk { uint32_t type, n, cnt, pbits; };  // per (tile, slot): block type 0..4, symbols, emitted words, piece bits
struct M2Tile {                                   // per tile, filled by k_m2_select
    uint32_t kind;     // 0 raw colour, 1 colour, 2 gray, 3 raw gray, 4 single colour
    uint32_t m;        // gray: chosen predictor
    uint32_t bbits;    // total bits of b (head + pieces)
    uint32_t size;     // blob bytes
};
constexpr uint32_t M2F_NOT_SINGLE = 1, M2F_NOT_GRAY = 2;

// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_m2_classify(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                     const TileDesc *__restrict__ tiles, TileSel sel, uint32_t strips,
                                                     uint32_t *__restrict__ flags) {
    const uint32_t tile = vtile(sel, blockIdx.x / strips), strip = blockIdx.x % strips;
    const TileDesc t = tiles[tile];
    const uint8_t *raster = rasters[t.img];
    const uint8_t *base = raster + (uint64_t)t.y * bpr + (uint64_t)t.x * 3;
    const uint32_t first = load_px<3>(base);
    const uint32_t i0 = (uint32_t)((uint64_t)t.n * strip / strips), i1 = (uint32_t)((uint64_t)t.n * (strip + 1) / strips);
    uint32_t f = 0;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const uint32_t y = i / t.w, x = i - y * t.w;
        const uint32_t p = load_px<3>(base + (uint64_t)y * bpr + (uint64_t)x * 3);
        if (p != first) f |= M2F_NOT_SINGLE;
        if (((p >> 8) & 0xFFFF) != (p & 0xFFFF)) f |= M2F_NOT_GRAY;  // g,b == r,g  <=>  r == g == b
    }
    const uint64_t a = __ballot(f & 1), b = __ballot(f & 2);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = (a ? M2F_NOT_SINGLE : 0u) | (b ? M2F_NOT_GRAY : 0u);
        if (w) atomicOr(&flags[tile], w);
    }
}

// --------------------------------------------------------------------------------------------------
// stream lengths of every tile, before anything is routed (the layout above needs them): a colour tile's nine context streams as
// in k_m1_lens, its class stream v holds one symbol (v = 1, 2) or three (v >= 3) per coded pixel with nl = v; a gray tile's four
// candidate streams hold n - 1 symbols each; everything else 0.  The histogram of the nl plane comes from the transform itself
// (nlh, m1_encode.hpp [r4]: rounds 2-3 re-read the plane here).   grid = tiles, block = 64.
__global__ __launch_bounds__(64) void k_m2_count(const TileDesc *__restrict__ tiles, TileSel sel, const uint32_t *__restrict__ flags,
                                                 const uint32_t *__restrict__ nlh, uint32_t slots, uint32_t generic, uint32_t *__restrict__ stream_n) {
    const uint32_t tile = vtile(sel, blockIdx.x), tid = threadIdx.x;
    const TileDesc t = tiles[tile];
    const uint32_t f = flags[tile];
    uint32_t *sn = stream_n + (uint64_t)tile * M2_SLOTS;
    if ((f & (M2F_NOT_SINGLE | M2F_NOT_GRAY)) != (M2F_NOT_SINGLE | M2F_NOT_GRAY)) {  // single colour or gray
        const bool gray = (f & M2F_NOT_SINGLE) && !(f & M2F_NOT_GRAY);
        if (tid < M2_SLOTS) sn[tid] = gray && tid >= 17 ? t.n - 1 : 0u;
        return;
    }
    uint32_t h[10];
    nlh_reduce(nlh + (uint64_t)tile * slots * NLH_STRIDE, nlh_records(t.w, t.h, generic != 0), h);
    if (tid < 9) {
        uint32_t hc = 0;
#pragma unroll
        for (int c = 0; c < 9; c++) hc = tid == (uint32_t)c ? h[c] : hc;
        uint32_t len = hc;
        if (h[9]) len = len - ((h[9] & 15u) == tid ? 1u : 0u) + (tid == 0 ? 1u : 0u);
        sn[tid] = len;
        if (tid >= 1) sn[8 + tid] = hc * (tid >= 3 ? 3u : 1u);
    }
    if (tid >= 17 && tid < M2_SLOTS) sn[tid] = 0;
}

// --------------------------------------------------------------------------------------------------
// routing for colour tiles: context streams by pl (as k_m1_streams) and class streams by nl.  grid = tiles, block = THREADS
// (1024 for one image: the tile's latency; 256 for batches).  A lane owns FOUR consecutive pixels (one dword of each plane, read an
// iteration ahead); the append position of a pixel in its context stream and in its class stream are prefix sums of nine and
// eight counters, packed three 10-bit fields to a word: six DPP scans per 4 x THREADS pixels (rounds 1-3: one pixel per lane and
// seventeen ballots per THREADS pixels).  Three barriers per iteration.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_m2_streams(const TileDesc *__restrict__ tiles, TileSel sel, const uint32_t *__restrict__ flags,
                                                     const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                     uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                     const uint32_t *__restrict__ stream_n) {
    const uint32_t tile = vtile(sel, blockIdx.x);
    const TileDesc t = tiles[tile];
    if ((flags[tile] & (M2F_NOT_SINGLE | M2F_NOT_GRAY)) != (M2F_NOT_SINGLE | M2F_NOT_GRAY)) return;  // single colour or gray
    constexpr int WAVES = THREADS / 64, PX = THREADS * 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t *pnl = reinterpret_cast<const uint32_t *>(planes + t.pbase), *pr_ = reinterpret_cast<const uint32_t *>(planes + plane_stride + t.pbase);
    const uint32_t *pg = reinterpret_cast<const uint32_t *>(planes + 2 * plane_stride + t.pbase), *pb = reinterpret_cast<const uint32_t *>(planes + 3 * plane_stride + t.pbase);
    uint8_t *sc = scratch2 + sbase2[tile];
    __shared__ uint32_t s_off[17];                 // places of the 17 streams (their lengths are known: k_m2_count)
    __shared__ uint32_t s_run_ctx[9], s_run_cls[9];  // symbols routed so far: per context stream; per class, in PIXELS
    __shared__ uint32_t s_wave_ctx[WAVES][9], s_wave_cls[WAVES][9];
    __shared__ volatile uint32_t s_bctx[WAVES][16], s_bcls[WAVES][16];
    __shared__ uint32_t s_wave_last[WAVES];
    if (tid < 9) { s_run_ctx[tid] = 0; s_run_cls[tid] = 0; }
    if (tid < 17) s_off[tid] = (uint32_t)m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, tid);
    __syncthreads();
    uint32_t run_pl = 0;
    const uint64_t lt = lanemask_lt();
    // field of counter c in the packed words: word c / 3, bits [10 * (c % 3), +10)
    const uint32_t myq = __umul24(lane, 11u) >> 5, mysh = __umul24(lane - __umul24(myq, 3u), 10u);
    uint32_t nx_nl = 0xFFFFFFFFu, nx_r = 0, nx_g = 0, nx_b = 0;
    if (4 * tid < t.n) { nx_nl = pnl[tid]; nx_r = pr_[tid]; nx_g = pg[tid]; nx_b = pb[tid]; }
    for (uint32_t i0 = 0; i0 < t.n; i0 += PX) {
        const uint32_t i = i0 + 4 * tid;
        uint32_t nl4 = nx_nl;
        const uint32_t r4 = nx_r, g4 = nx_g, b4 = nx_b;
        {
            const uint32_t in = i + PX;
            nx_nl = 0xFFFFFFFFu;
            if (in < t.n) { nx_nl = pnl[in >> 2]; nx_r = pr_[in >> 2]; nx_g = pg[in >> 2]; nx_b = pb[in >> 2]; }
        }
        if (i < t.n && t.n - i < 4) nl4 |= 0xFFFFFFFFu << (8 * (t.n - i));  // pixels past the tile: not coded
        uint32_t nl[4], csh[4], cinc[4][3], c0 = 0, c1 = 0, c2 = 0;
        bool coded[4];
        uint32_t lastnl = NL_NONE;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            nl[j] = (nl4 >> (8 * j)) & 255u;
            coded[j] = nl[j] != NL_NONE;
            lastnl = coded[j] ? nl[j] : lastnl;
            // class counter of the pixel: field nl (1..8; nl = 0 emits no class symbol)
            const uint32_t v = coded[j] ? (nl[j] & 15u) : 0u, q = __umul24(v, 11u) >> 5;
            csh[j] = __umul24(v - __umul24(q, 3u), 10u);
            const uint32_t one = v ? 1u << csh[j] : 0u;
            cinc[j][0] = q == 0 ? one : 0u; cinc[j][1] = q == 1 ? one : 0u; cinc[j][2] = q == 2 ? one : 0u;
            c0 += cinc[j][0]; c1 += cinc[j][1]; c2 += cinc[j][2];
        }
        const uint64_t mask = __ballot(nl4 != 0xFFFFFFFFu);
        const uint64_t lower = mask & lt;
        const uint32_t wave_last = __shfl(lastnl, mask ? 63 - __clzll((long long)mask) : 0);
        if (lane == 0) s_wave_last[wv] = mask ? wave_last : NL_NONE;
        const uint32_t prev_in_wave = __shfl(lastnl, lower ? 63 - __clzll((long long)lower) : 0);
        const uint32_t y0 = wave_scan_incl(c0), y1 = wave_scan_incl(c1), y2 = wave_scan_incl(c2);
        {
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)y0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)y1, 63),
                           t2 = (uint32_t)__builtin_amdgcn_readlane((int)y2, 63);
            const uint32_t word = myq == 0 ? t0 : (myq == 1 ? t1 : t2);
            if (lane < 9) s_wave_cls[wv][lane] = (word >> mysh) & 1023u;
        }
        __syncthreads();  // (A) wave_last / class counts visible
        uint32_t carry = run_pl, new_run_pl = run_pl;
        for (int w2 = 0; w2 < WAVES; w2++) {
            const uint32_t wl = s_wave_last[w2];
            if (w2 < (int)wv && wl != NL_NONE) carry = wl;
            if (wl != NL_NONE) new_run_pl = wl;
        }
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
            if (lane < 9) s_wave_ctx[wv][lane] = (word >> mysh) & 1023u;
        }
        __syncthreads();  // (B) per-wave context counts visible
        if (lane < 9) {   // this wave's base position in each context stream (bytes inside the tile's scratch) and in each class (pixels)
            uint32_t bc = s_off[lane] + s_run_ctx[lane], bk = s_run_cls[lane];
            for (uint32_t w = 0; w < wv; w++) { bc += s_wave_ctx[w][lane]; bk += s_wave_cls[w][lane]; }
            s_bctx[wv][lane] = bc; s_bcls[wv][lane] = bk;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t e0 = x0 - w0, e1 = x1 - w1, e2 = x2 - w2s;  // exclusive lane prefixes, advanced pixel by pixel
        uint32_t f0 = y0 - c0, f1 = y1 - c1, f2 = y2 - c2;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (coded[j]) {
                const uint32_t e = inc[j][0] ? e0 : (inc[j][1] ? e1 : e2);
                sc[s_bctx[wv][pl[j]] + ((e >> sh[j]) & 1023u)] = (uint8_t)nl[j];
                const uint32_t v = nl[j];
                if (v) {
                    const uint32_t f = cinc[j][0] ? f0 : (cinc[j][1] ? f1 : f2);
                    const uint32_t k = s_bcls[wv][v] + ((f >> csh[j]) & 1023u);
                    uint8_t *dst = sc + s_off[8 + v];
                    const uint32_t zr = (r4 >> (8 * j)) & 255u, zg = (g4 >> (8 * j)) & 255u, zb = (b4 >> (8 * j)) & 255u;
                    if (v == 1) dst[k] = (uint8_t)((zr << 2) | (zg << 1) | zb);        // libxpng.c:37
                    else if (v == 2) dst[k] = (uint8_t)((zr << 4) | (zg << 2) | zb);   // libxpng.c:38
                    else { uint8_t *q = dst + 3 * k; q[0] = (uint8_t)zr; q[1] = (uint8_t)zg; q[2] = (uint8_t)zb; }  // :39
                }
            }
            e0 += inc[j][0]; e1 += inc[j][1]; e2 += inc[j][2];
            f0 += cinc[j][0]; f1 += cinc[j][1]; f2 += cinc[j][2];
        }
        uint32_t tot_ctx = 0, tot_cls = 0;
        if (tid < 9) for (int w2 = 0; w2 < WAVES; w2++) { tot_ctx += s_wave_ctx[w2][tid]; tot_cls += s_wave_cls[w2][tid]; }
        __syncthreads();  // (C) every wave has read the running counts and the per-wave counts
        if (tid < 9) { s_run_ctx[tid] += tot_ctx; s_run_cls[tid] += tot_cls; }
        run_pl = new_run_pl;
        // (the next iteration's barrier (A) orders these LDS writes before their next use)
    }
}

// --------------------------------------------------------------------------------------------------
constexpr uint32_t M2_GRAY_REPS = 16;
// gray tiles: the four candidate symbol streams (libxpng.c:597-604).  grid = tiles * blocks_per_tile, block = 256.
__global__ __launch_bounds__(256) void k_m2_gray_syms(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                      const TileDesc *__restrict__ tiles, TileSel sel, uint32_t bpt,
                                                      const uint32_t *__restrict__ flags, uint8_t *__restrict__ scratch2,
                                                      const uint64_t *__restrict__ sbase2, const uint32_t *__restrict__ stream_n) {
    const uint32_t tile = vtile(sel, blockIdx.x / bpt), chunk = blockIdx.x % bpt;
    const TileDesc t = tiles[tile];
    const uint32_t f = flags[tile];
    if (!(f & M2F_NOT_SINGLE) || (f & M2F_NOT_GRAY)) return;  // only gray, not single-colour, tiles
    const uint8_t *raster = rasters[t.img];
    uint8_t *sc = scratch2 + sbase2[tile];
    const uint64_t o17 = m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, 17), gs = m2_slot(t.n - 1);  // (k_m2_count set the four lengths to n - 1)
    for (uint32_t rep = 0; rep < M2_GRAY_REPS; rep++) {  // (4096 pixels per workgroup: a colour image launches these to find nothing)
    const uint32_t i = (chunk * M2_GRAY_REPS + rep) * 256 + threadIdx.x + 1;  // pixel 1..n-1 -> symbol i-1
    if (i >= t.n) return;
    const uint32_t y = i / t.w, x = i - y * t.w;
    const uint8_t *p = raster + (uint64_t)(t.y + y) * bpr + (uint64_t)(t.x + x) * 3;
    const int v = p[0];
    int s0, s1, s2, s3;
    if (y == 0) s0 = s1 = s2 = s3 = zz_enc(v - p[-3]);
    else if (x == 0) s0 = s1 = s2 = s3 = zz_enc(v - *(p - bpr));
    else {
        const int L = p[-3], U = *(p - bpr), UL = *(p - bpr - 3);
        s0 = zz_enc(v - L); s1 = zz_enc(v - U); s2 = zz_enc(v - pred_avg(L, U)); s3 = zz_enc(v - pred_grad(L, U, UL));
    }
    sc[o17 + i - 1] = (uint8_t)s0;
    sc[o17 + gs + i - 1] = (uint8_t)s1;
    sc[o17 + 2 * gs + i - 1] = (uint8_t)s2;
    sc[o17 + 3 * gs + i - 1] = (uint8_t)s3;
    }
}

// --------------------------------------------------------------------------------------------------
// rANS v1 block (compress_block, libxpng.c:160-260).  One wave per (tile, slot).  The block slot receives
// [state0 lo,hi][state1 lo,hi][words in EMISSION order]; k_m2_gather writes the final block with the words reversed.
__global__ __launch_bounds__(64) void k_rans1_encode(const TileDesc *__restrict__ tiles, TileSel sel, const uint32_t *__restrict__ flags,
                                                     uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                     const uint32_t *__restrict__ stream_n, M2Blk *__restrict__ blk) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cum[260];
    __shared__ EncSym tab[256];
    __shared__ uint32_t ring[512];
    __shared__ __align__(16) uint8_t sring[2048];
    const uint32_t tile = vtile(sel, blockIdx.x / M2_SLOTS), slot = blockIdx.x % M2_SLOTS, lane = threadIdx.x & 63, par = lane & 1;
    const TileDesc t = tiles[tile];
    const uint32_t f = flags[tile];
    const bool single = !(f & M2F_NOT_SINGLE), gray = !single && !(f & M2F_NOT_GRAY);
    if (single || (gray != (slot >= 17))) return;  // colour tiles run slots 0..16, gray tiles 17..20
    uint8_t *sc = scratch2 + sbase2[tile];
    const uint8_t *in = sc + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot);
    const uint32_t n = sgpr(stream_n[(uint64_t)tile * M2_SLOTS + slot]);
    const uint32_t Nnom = m2_nominal(slot);
    const int pb = slot >= 17 ? 15 : 14;
    uint32_t *out = reinterpret_cast<uint32_t *>(sc + m2_off_blk(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot));
    M2Blk *mb = blk + (uint64_t)tile * M2_SLOTS + slot;
    if (n == 0) {  // libxpng.c:167
        if (lane == 0) *mb = M2Blk{0, 0, 0, 0};
        return;
    }
    rans_histogram(in, n, hist);
    uint32_t top, distinct;
    rans_alphabet(hist, Nnom, top, distinct);
    if (distinct == 1) {  // libxpng.c:169-172: the block carries the symbol
        if (lane == 0) *mb = M2Blk{1, n, (uint32_t)in[n - 1], 0};
        return;
    }
    const uint32_t N = top + 1;
    rans_tables(hist, cum, tab, N, n, pb);
    // ---- recurrence, backwards (libxpng.c:215-243).  even lanes: state0 / even symbol indices; odd lanes: state1.
    const uint32_t cmpl_base = 1u << pb;
    const int thr_shift = 31 - pb;
    uint64_t s = RANS_L;
    uint32_t cnt = 0, flushed = 0;
    uint32_t *w = out + 4;
    // The recurrence runs BACKWARDS over the symbols; otherwise the structure of the v2 encoder (rans2_encode_block): straight-line
    // double blocks of 8 steps, the 8 symbols of a block = one unaligned 8-byte LDS read issued two blocks ahead, its four table
    // entries read one block ahead, the spill handling behind a wave-uniform branch, ring flush and unit rotation checked once
    // per 128 steps.  The symbol ring (2 KB, 1 KB units, the stream's 16-byte phase kept) is filled downwards; its first 16
    // bytes are mirrored behind its end so that a window read never wraps.
    const uint8_t *inA = reinterpret_cast<const uint8_t *>((uintptr_t)in & ~(uintptr_t)15);
    const uint32_t p0 = sgpr((uint32_t)((uintptr_t)in & 15));
    const uint4 *src = reinterpret_cast<const uint4 *>(inA) + lane;
    __shared__ __align__(16) uint8_t smirror[16];
    static_assert(sizeof(sring) == 2048, "ring size");
    auto put_unit = [&](int32_t unit, const uint4 &v) __attribute__((always_inline)) {
        reinterpret_cast<uint4 *>(sring)[(unit & 1) * 64 + lane] = v;
        if ((unit & 1) == 0 && lane == 0) *reinterpret_cast<uint4 *>(smirror) = v;
    };
    int32_t lo_unit = (int32_t)sgpr((p0 + n - 1) >> 10);  // lowest unit staged; units lo_unit and lo_unit+1 (if it exists) are in the ring
    put_unit(lo_unit, src[lo_unit * 64]);
    if (lo_unit > 0) { lo_unit--; put_unit(lo_unit, src[lo_unit * 64]); }
    uint4 pre = lo_unit > 0 ? src[(lo_unit - 1) * 64] : make_uint4(0, 0, 0, 0);
    __syncthreads();
    auto put = [&](const EncSym &e) __attribute__((always_inline)) {
        const uint32_t freq = e.freq_shift & 0xFFFF, rsh = e.freq_shift >> 16;
        const uint64_t rcp = ((uint64_t)e.rcp_hi << 32) | e.rcp_lo;
        const uint64_t q = __umul64hi(s, rcp) >> rsh;
        s += e.bias + q * (uint64_t)(cmpl_base - freq);
    };
    const uint32_t pairs = sgpr(n >> 1);
    if (n & 1) { if (!par) put(tab[in[n - 1]]); }  // libxpng.c:218-225: no spill test on the odd tail
    auto step = [&](const EncSym &e) __attribute__((always_inline)) {
        const uint32_t freq = e.freq_shift & 0xFFFF;
        const bool emit = (uint32_t)(s >> 32) >= (freq << thr_shift);
        const uint32_t m = sgpr((uint32_t)__ballot(emit) & 3u);  // bit1: state1 spills (first), bit0: state0
        if (m) {  // uniform
            const uint32_t e1 = m >> 1;
            if (emit) {
                if (lane < 2) ring[(cnt + (par ? 0u : e1)) & 511u] = (uint32_t)s;
                s >>= 32;
            }
            cnt += e1 + (m & 1u);
        }
        put(e);
    };
    auto flush256 = [&]() __attribute__((always_inline)) {
        __syncthreads();
        for (uint32_t i = lane; i < 256; i += 64) w[flushed + i] = ring[(flushed + i) & 511u];
        flushed += 256;
        __syncthreads();
    };
    auto rotate = [&]() __attribute__((always_inline)) {  // bring the prefetched lower unit in
        __syncthreads();
        lo_unit--;
        put_unit(lo_unit, pre);
        if (lo_unit > 0) pre = src[(lo_unit - 1) * 64];
        __syncthreads();
    };
    // (the mirror lives right behind the ring: one contiguous 2064-byte object for the unaligned window reads)
    typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
    auto window = [&](int32_t pos) -> u32x2_a1 {  // ring bytes [pos, pos + 8)
        const uint32_t a = (uint32_t)pos & 2047u;
        const u32x2_a1 v = *reinterpret_cast<const u32x2_a1 *>(sring + a);
        if (a <= 2040u) return v;
        // the read crossed the end of the ring: splice in the mirror of its first bytes (rare: 7 of 2048 positions)
        uint8_t tmp[8];
#pragma unroll
        for (int q = 0; q < 8; q++) tmp[q] = (a + q) < 2048u ? sring[a + q] : smirror[a + q - 2048u];
        u32x2_a1 r;
        r.x = tmp[0] | tmp[1] << 8 | tmp[2] << 16 | (uint32_t)tmp[3] << 24; r.y = tmp[4] | tmp[5] << 8 | tmp[6] << 16 | (uint32_t)tmp[7] << 24;
        return r;
    };
    const uint32_t bsh = 8u * par;
    // step u of a block (u = 0..3, downwards) codes the pair whose even symbol is byte 6 - 2u of the block's window
    auto entry = [&](const u32x2_a1 &Wn, int u) -> EncSym {
        const uint32_t d = u < 2 ? Wn.y : Wn.x;
        const uint32_t sy = (d >> (bsh + 16u * (uint32_t)(1 - (u & 1)))) & 255u;
        return tab[sy];
    };
    int32_t pos = (int32_t)(p0 + 2 * pairs) - 2;  // ring position of the even symbol of the next pair (uniform); may run below p0
    const uint32_t n8 = pairs >> 3;
    uint32_t db = 0;
    if (n8) {
        // block A covers ring bytes [pos - 6, pos + 2), block B the eight below, ...
        u32x2_a1 Wa = window(pos - 6), Wb = window(pos - 14);
        EncSym Ea[4], Eb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) Ea[u] = entry(Wa, u);
        while (db < n8) {
            if (cnt - flushed >= 256) flush256();
            if (lo_unit > 0 && pos - 320 < lo_unit * 1024) rotate();
            const uint32_t stop = sgpr(db + 16 < n8 ? db + 16 : n8);
#pragma unroll 1
            for (; db < stop; db++) {
                Wa = window(pos - 22);
#pragma unroll
                for (int u = 0; u < 4; u++) Eb[u] = entry(Wb, u);
#pragma unroll
                for (int u = 0; u < 4; u++) step(Ea[u]);
                Wb = window(pos - 30);
#pragma unroll
                for (int u = 0; u < 4; u++) Ea[u] = entry(Wa, u);
#pragma unroll
                for (int u = 0; u < 4; u++) step(Eb[u]);
                pos -= 16;
            }
        }
    }
    // ---- tail: fewer than 8 pair steps, one at a time straight from the ring
    __syncthreads();
    if (cnt - flushed >= 256) flush256();
    if (lo_unit > 0 && pos - 64 < lo_unit * 1024) rotate();
    for (uint32_t k = n8 * 8; k < pairs; k++) {
        step(tab[sring[(uint32_t)(pos + (int32_t)par) & 2047u]]);
        pos -= 2;
    }
    __syncthreads();
    for (uint32_t i = flushed + lane; i < cnt; i += 64) w[i] = ring[i & 511u];
    if (lane < 2) { out[2 * lane] = (uint32_t)s; out[2 * lane + 1] = (uint32_t)(s >> 32); }  // state0 then state1 (libxpng.c:245)
    // ---- type decision and the table piece (libxpng.c:247-259)
    const uint32_t rawBits = (uint32_t)bit_width(Nnom - 1);
    uint32_t tabBits = (Nnom - distinct) + distinct * ((uint32_t)pb + 1);
    const bool sparse = tabBits < Nnom * (uint32_t)pb;
    if (!sparse) tabBits = Nnom * (uint32_t)pb;
    const uint64_t ransBytes = 16 + 4ull * cnt;
    if ((uint64_t)tabBits + 8 * ransBytes >= (uint64_t)rawBits * n) {
        if (lane == 0) *mb = M2Blk{2, n, 0, rawBits * n};  // raw symbols travel in b
        return;
    }
    if (lane == 0) {
        BitW tb{0, 0, reinterpret_cast<uint32_t *>(sc + m2_off_piece(t.n, slot))};
        for (uint32_t k = 0; k < Nnom; k++) {
            const uint32_t F = k < N ? hist[k] : 0;
            if (!sparse) tb.put((uint32_t)pb, F);
            else if (F) tb.put((uint32_t)pb + 1, F + (1u << pb));
            else tb.put(1, 0);
        }
        tb.finish();
        *tb.p = 0;
        *mb = M2Blk{3u + (sparse ? 1u : 0u), n, cnt, tabBits};
    }
}

__device__ __forceinline__ uint32_t m2_blk_bytes(const M2Blk &b) {  // final block size
    return b.type == 0 ? 4u : (b.type <= 2 ? 8u : 24u + 4u * b.cnt);
}

// --------------------------------------------------------------------------------------------------
// tile kind, gray choice, sizes.  One thread per tile.
__global__ void k_m2_select(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t total, const uint32_t *__restrict__ flags,
                            const M2Blk *__restrict__ blk, M2Tile *__restrict__ mt, uint32_t *__restrict__ tile_sz) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const uint32_t f = flags[tile];
    const M2Blk *b = blk + (uint64_t)tile * M2_SLOTS;
    M2Tile r{};
    if (!(f & M2F_NOT_SINGLE)) { r.kind = 4; r.size = 8; }
    else if (!(f & M2F_NOT_GRAY)) {
        uint64_t bestB = 1000000, bestR = 1000000;  // libxpng.c:606
        for (uint32_t m = 0; m < 4; m++) {
            const uint64_t bsz = 4 + 4ull * ((8 + b[17 + m].pbits + 31) >> 5), rsz = m2_blk_bytes(b[17 + m]);
            if (bsz + rsz < bestB + bestR) { r.m = m; bestB = bsz; bestR = rsz; }
        }
        if (bestB + bestR >= t.n) { r.kind = 3; r.size = t.n + 4; }
        else { r.kind = 2; r.bbits = 8 + b[17 + r.m].pbits; r.size = (uint32_t)(bestB + bestR + 4); }
    } else {
        uint64_t bits = 24, rsz = 0;
        for (uint32_t s = 0; s < M2_STREAMS; s++) { bits += b[s].pbits; rsz += m2_blk_bytes(b[s]); }
        const uint64_t bsz = 4 + 4 * ((bits + 31) >> 5);
        if (bsz + rsz >= 3ull * t.n) { r.kind = 0; r.size = 3 * t.n + 4; }  // libxpng.c:675
        else { r.kind = 1; r.bbits = (uint32_t)bits; r.size = (uint32_t)(bsz + rsz + 4); }
    }
    mt[tile] = r;
    tile_sz[j] = r.size;
}

// --------------------------------------------------------------------------------------------------
// shared bit stream b: head (first pixel) + pieces at arbitrary bit offsets.  One thread per output word.
// grid = tiles, block = 256.
__global__ __launch_bounds__(256) void k_m2_bits(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                 const TileDesc *__restrict__ tiles, TileSel sel, const M2Tile *__restrict__ mt,
                                                 const M2Blk *__restrict__ blk, uint8_t *__restrict__ scratch2,
                                                 const uint64_t *__restrict__ sbase2, const uint32_t *__restrict__ stream_n) {
    const uint32_t tile = vtile(sel, blockIdx.x);
    const TileDesc t = tiles[tile];
    const M2Tile r = mt[tile];
    if (r.kind != 1 && r.kind != 2) return;
    const uint8_t *p0 = rasters[t.img] + (uint64_t)t.y * bpr + (uint64_t)t.x * 3;
    uint8_t *sc = scratch2 + sbase2[tile];
    const M2Blk *b = blk + (uint64_t)tile * M2_SLOTS;
    __shared__ uint32_t s_off[19], s_slot[18];
    __shared__ uint32_t s_np;
    if (threadIdx.x == 0) {
        uint32_t o, np = 0;
        if (r.kind == 2) { o = 8; s_off[0] = o; s_slot[0] = 17 + r.m; o += b[17 + r.m].pbits; np = 1; }
        else { o = 24; for (uint32_t s = 0; s < M2_STREAMS; s++) { s_off[np] = o; s_slot[np] = s; o += b[s].pbits; np++; } }
        s_off[np] = o;
        s_np = np;
    }
    __syncthreads();
    const uint32_t np = s_np, words = (r.bbits + 31) >> 5;
    uint32_t *out = reinterpret_cast<uint32_t *>(sc + m2_off_bits(t.n));
    const uint32_t head = r.kind == 2 ? ((uint32_t)p0[0] << 24) : (((uint32_t)p0[0] << 24) | ((uint32_t)p0[1] << 16) | ((uint32_t)p0[2] << 8));
    for (uint32_t wi = threadIdx.x; wi < words; wi += 256) {
        const uint32_t w0 = wi * 32, w1 = w0 + 32;
        uint32_t val = wi == 0 ? head : 0u;
        for (uint32_t k = 0; k < np; k++) {
            const uint32_t o0 = s_off[k], o1 = s_off[k + 1];
            if (o1 <= w0 || o0 >= w1 || o0 == o1) continue;
            const uint32_t a = o0 > w0 ? o0 : w0, e = o1 < w1 ? o1 : w1;  // overlap [a, e) in stream bits
            const uint32_t slot = s_slot[k];
            const M2Blk mb = b[slot];
            if (mb.type == 2) {  // raw symbols, rawBits each, MSB first (libxpng.c:251)
                const uint32_t rb = mb.pbits / mb.n;
                const uint8_t *st = sc + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot);
                uint32_t jsym = (a - o0) / rb;
                for (;; jsym++) {
                    const uint32_t s0 = o0 + jsym * rb, s1 = s0 + rb;  // this symbol's bits in the stream
                    if (s0 >= e) break;
                    const uint32_t lo = s0 > a ? s0 : a, hi = s1 < e ? s1 : e;
                    const uint32_t v = ((uint32_t)st[jsym] >> (s1 - hi)) & ((1u << (hi - lo)) - 1);
                    val |= v << (w1 - hi);
                }
            } else {  // frequency table bits from the piece buffer (bit 0 = MSB of its word 0)
                const uint32_t *pw = reinterpret_cast<const uint32_t *>(sc + m2_off_piece(t.n, slot));
                const uint32_t rel = a - o0, c = e - a;
                const uint64_t two = ((uint64_t)pw[rel >> 5] << 32) | pw[(rel >> 5) + 1];
                const uint32_t v = (uint32_t)((two >> (64 - (rel & 31) - c)) & (c == 32 ? 0xFFFFFFFFull : ((1ull << c) - 1)));
                val |= c == 32 ? v : (v << (w1 - e));
            }
        }
        out[wi] = val;
    }
}

// --------------------------------------------------------------------------------------------------
// final placement.  grid = tiles, block = 256.
__device__ inline void m2_write_block(uint8_t *dst, const M2Blk &mb, const uint32_t *slotw) {
    auto w32 = [&](uint32_t byteoff, uint32_t v) { for (int k = 0; k < 4; k++) dst[byteoff + k] = (uint8_t)(v >> (8 * k)); };
    if (mb.type == 0) { if (threadIdx.x == 0) w32(0, 4); return; }
    if (mb.type == 1) { if (threadIdx.x == 0) { w32(0, 8u | (1u << 24)); w32(4, mb.n + (mb.cnt << 24)); } return; }
    if (mb.type == 2) { if (threadIdx.x == 0) { w32(0, 8u | (2u << 24)); w32(4, mb.n); } return; }
    if (threadIdx.x == 0) { w32(0, (24u + 4u * mb.cnt) | (mb.type << 24)); w32(4, mb.n); }
    for (uint32_t i = threadIdx.x; i < 4 + mb.cnt; i += blockDim.x)  // states, then the words in reverse emission order
        w32(8 + 4 * i, i < 4 ? slotw[i] : slotw[4 + (mb.cnt - 1 - (i - 4))]);
}

__global__ __launch_bounds__(256) void k_m2_gather(const uint8_t *const *__restrict__ rasters, uint64_t bpr,
                                                   const TileDesc *__restrict__ tiles, TileSel sel, const uint32_t *__restrict__ sums,
                                                   const M2Tile *__restrict__ mt, const M2Blk *__restrict__ blk,
                                                   const uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                   const uint64_t *__restrict__ off, uint8_t *const *__restrict__ blobs) {
    const uint32_t j = blockIdx.x, tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const M2Tile r = mt[tile];
    const uint8_t *raster = rasters[t.img];
    const uint8_t *src = raster + (uint64_t)t.y * bpr + (uint64_t)t.x * 3;
    uint8_t *dst = blobs[t.img] + off[(uint64_t)(j / sel.cnt) * (sel.cnt + 1) + j % sel.cnt];
    const uint8_t *sc = scratch2 + sbase2[tile];
    const M2Blk *b = blk + (uint64_t)tile * M2_SLOTS;
    auto w32 = [&](uint32_t byteoff, uint32_t v) { for (int k = 0; k < 4; k++) dst[byteoff + k] = (uint8_t)(v >> (8 * k)); };
    if (r.kind == 4) {  // single colour (libxpng.c:637-640)
        if (threadIdx.x == 0) { w32(0, (255u << 24) | 8u); dst[4] = src[0]; dst[5] = src[1]; dst[6] = src[2]; dst[7] = 0; }
        return;
    }
    if (r.kind == 0) {  // raw rows (libxpng.c:675-677)
        if (threadIdx.x == 0) w32(0, r.size);
        const uint64_t row = (uint64_t)t.w * 3;
        for (uint32_t y = 0; y < t.h; y++) block_copy(dst + 4 + y * row, src + y * bpr, row);
        return;
    }
    if (r.kind == 3) {  // raw gray (libxpng.c:616-618)
        if (threadIdx.x == 0) w32(0, (t.n + 4) + (5u << 27));
        for (uint32_t i = threadIdx.x; i < t.n; i += 256) { const uint32_t y = i / t.w, x = i - y * t.w; dst[4 + i] = src[(uint64_t)y * bpr + (uint64_t)x * 3]; }
        return;
    }
    const uint32_t bsz = 4 + 4 * ((r.bbits + 31) >> 5);
    if (threadIdx.x == 0) {
        if (r.kind == 2) w32(0, r.size + (2u << 28) + (r.m << 24));
        else w32(0, r.size + (1u << 28) + (((uint32_t)pr_from_sums(sums + (uint64_t)tile * 4, 3, t.w, t.h) & 3u) << 24));
        w32(4, bsz);
    }
    block_copy(dst + 8, sc + m2_off_bits(t.n), bsz - 4);
    uint32_t o = 4 + bsz;
    if (r.kind == 2) {
        const uint32_t s = 17 + r.m;
        m2_write_block(dst + o, b[s], reinterpret_cast<const uint32_t *>(sc + m2_off_blk(t.n, nullptr, s)));
    } else {
        uint64_t bo = m2_off_blk(t.n, nullptr, 0);  // block slots back to back, each sized by its stream's length (= M2Blk::n)
        for (uint32_t s = 0; s < M2_STREAMS; s++) {
            m2_write_block(dst + o, b[s], reinterpret_cast<const uint32_t *>(sc + bo));
            o += m2_blk_bytes(b[s]);
            bo += m2_blk_cap(b[s].n);
        }
    }
}

}  // namespace xpng
