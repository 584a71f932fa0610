// m2_decode.hpp -- mode-2 (RGB slow level) tile DECODE kernels for gfx950.
//
// Reference path restated: dec_2_th (libxpng.c:929-961) = raw / when_single_color (916-927) / when_grayscale (868-899) /
// 17x decompress_block (rANS v1, 262-301) + DEC/DEC4 (901-914).  Stages:
//
//   k_m2_dec_parse   one thread per tile: tile kind, the 17 (or 1) block headers, and the frequency tables, which sit in the
//                    shared bit stream `b` in block order and must be read serially (sparse tables have data-dependent length)
//   k_rans1_decode   one wave per (tile, stream): forwards over the symbols, state0 refills before state1; same hot-symbol
//                    cache / LDS tables / scalar word cursor as the v2 decoder
//   ctx_walk         the nl context chain (shared with mode 1)
//   k_m2_dec_resid   class-stream routing back to pixels (rank by nl: packed prefix sums, four pixels per lane), zig-zag / green add-back
//   k_m2_dec_recon   fill / raw copy / anti-diagonal wavefront (shared with mode 1); gray tiles use predictor m
#pragma once
#include "common.hpp"
#include "m1_decode.hpp"
#include "m2_encode.hpp"

namespace xpng {

struct M2DecTile {
    const uint8_t *blob;
    uint32_t kind;    // 0 raw colour, 1 colour, 2 gray, 3 raw gray, 4 single colour
    uint32_t m;       // colour: Y<<1|G ; gray: predictor
    uint32_t bsz;     // bytes of the b region including its size word
    uint32_t coded;   // colour: number of context symbols (= n - 1)
};
// M2Blk is reused per (tile, slot): type, n, cnt = byte offset of the block inside the blob, pbits = bit offset of its raw
// symbols in b (type 2) or its single symbol (type 1).

// random-access MSB-first bit read from the words at `w` (bits past `endbit` read as 0, like BITSTREAM_FILL past DP)
__device__ __forceinline__ uint32_t bits_at(const uint8_t *w, uint64_t pos, uint32_t c, uint64_t endbit) {
    const uint64_t wi = pos >> 5;
    const uint32_t a = wi * 32 < endbit ? ld32u(w + wi * 4) : 0u, b = (wi + 1) * 32 < endbit ? ld32u(w + wi * 4 + 4) : 0u;
    const uint64_t two = ((uint64_t)a << 32) | b;
    return (uint32_t)((two >> (64 - (pos & 31) - c)) & ((1ull << c) - 1));
}

constexpr uint32_t M2_KIND_BAD = 9;  // tile failed validation: every kernel skips it (see k_dec_parse in m1_decode.hpp)

__global__ void k_m2_dec_parse(const uint8_t *const *__restrict__ blobs, const uint64_t *__restrict__ off,
                               const uint64_t *__restrict__ blob_len, uint32_t cnt, uint32_t total,
                               const TileDesc *__restrict__ tiles, TileSel sel, M2DecTile *__restrict__ info,
                               M2Blk *__restrict__ blk, uint16_t *__restrict__ tabs, uint32_t *__restrict__ stream_n,
                               uint32_t *__restrict__ status) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    M2DecTile d{};
    d.blob = blobs[j / cnt] + off[j];
    const uint64_t avail = blob_len[j / cnt] > off[j] ? blob_len[j / cnt] - off[j] : 0;
    bool ok = avail >= 4;
    const uint32_t h0 = ok ? ld32u(d.blob) : 0, ty = h0 >> 24, L = h0 & 0xFFFFFF;
    ok = ok && L <= avail && L >= 8;
    M2Blk *mb = blk + (uint64_t)tile * M2_SLOTS;
    uint32_t *sn = stream_n + (uint64_t)tile * M2_SLOTS;  // symbol counts of the tile's streams: where the decoded streams go (m2_off_stream)
    for (uint32_t s = 0; s < M2_SLOTS; s++) sn[s] = 0;
    if (ty == 0) { d.kind = 0; ok = ok && L == 3 * t.n + 4; }
    else if (ty == 255) { d.kind = 4; ok = ok && L == 8; }
    else if ((ty >> 4) == 2) { d.kind = (ty & 8) ? 3 : 2; d.m = ty & 3; if (d.kind == 3) ok = ok && L == t.n + 4; }
    else if ((ty >> 4) == 1) { d.kind = 1; d.m = ty & 3; }
    else ok = false;
    if (ok && (d.kind == 1 || d.kind == 2)) {
        d.bsz = ld32u(d.blob + 4);
        ok = d.bsz >= 8 && (d.bsz & 3) == 0 && 4 + (uint64_t)d.bsz + 4 <= L;
    }
    if (ok && (d.kind == 1 || d.kind == 2)) {
        const uint8_t *bw = d.blob + 8;
        const uint64_t endbit = (uint64_t)(d.bsz - 4) * 8;
        uint64_t pos = d.kind == 1 ? 24 : 8;
        uint32_t o = 4 + d.bsz, coded = 0;
        uint64_t room = 0;  // bytes of the stream region the streams read so far take
        const uint32_t s0 = d.kind == 1 ? 0 : 17, s1 = d.kind == 1 ? M2_STREAMS : 18;
        for (uint32_t s = s0; s < s1 && ok; s++) {
            const uint32_t Nnom = m2_nominal(s), pb = s >= 17 ? 15 : 14, rawBits = (uint32_t)bit_width(Nnom - 1);
            ok = (uint64_t)o + 4 <= L;
            if (!ok) break;
            const uint32_t hdr = ld32u(d.blob + o), type = hdr >> 24, bsize = hdr & 0xFFFFFF;
            ok = type <= 4 && bsize >= (type == 0 ? 4u : 8u) && (uint64_t)o + bsize <= L && (type < 3 || bsize >= 24);
            if (!ok) break;
            const uint32_t w1 = type ? ld32u(d.blob + o + 4) : 0;
            const uint32_t ncap = (s >= 11 && s < 17) ? 3 * t.n : t.n;  // what the format allows a stream to hold
            const uint32_t nsym = type == 1 ? (w1 & 0xFFFFFF) : w1;
            room += m2_slot(type ? nsym : 0u);
            ok = nsym <= ncap && room <= m2_streams_region(t.n);  // (the streams lie back to back, each in the room its length needs)
            if (!ok) break;
            sn[s] = type ? nsym : 0u;
            M2Blk r{type, 0, o, 0};
            if (type == 1) { r.n = w1 & 0xFFFFFF; r.pbits = w1 >> 24; }
            else if (type == 2) { r.n = w1; r.pbits = (uint32_t)pos; pos += (uint64_t)w1 * rawBits; }
            else if (type == 3 || type == 4) {
                r.n = w1;
                uint16_t *F = tabs + ((uint64_t)tile * M2_SLOTS + s) * 256;
                for (uint32_t i = 0; i < Nnom; i++) {
                    uint32_t f;
                    if (type == 3) { f = bits_at(bw, pos, pb, endbit); pos += pb; }
                    else if (bits_at(bw, pos, 1, endbit)) { f = bits_at(bw, pos + 1, pb, endbit); pos += pb + 1; }
                    else { f = 0; pos += 1; }
                    F[i] = (uint16_t)f;
                }
            }
            if (s < 9) coded += r.n;
            mb[s] = r;
            o += hdr & 0xFFFFFF;
        }
        d.coded = coded;
        ok = ok && coded <= t.n - 1;
    }
    if (!ok) { d.kind = M2_KIND_BAD; atomicOr(status, 1u); }
    info[j] = d;
}

// --------------------------------------------------------------------------------------------------
// one v1 block -> symbols (decompress_block, libxpng.c:262-301).  Single-wave workgroup per (tile, slot).
template <int MAXPB>
__global__ __launch_bounds__(64) void k_rans1_decode(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                     uint32_t s_first, uint32_t s_count, const M2Blk *__restrict__ blk,
                                                     const uint16_t *__restrict__ tabs, uint8_t *__restrict__ scratch2,
                                                     const uint64_t *__restrict__ sbase2, const uint32_t *__restrict__ stream_n) {
    // per group of 8 slots: (F | cum << 16, symbol) of the symbol owning slot (g << 3): one LDS read resolves a cold slot whose
    // group lies inside one symbol's range, else a short forward scan over fc[] follows (as k_rans2_decode)
    __shared__ uint2 coarse[1 << (MAXPB - 3)];
    __shared__ uint32_t fc[256];
    __shared__ uint32_t wring[512];
    __shared__ __align__(8) uint8_t oring[512];
    const uint32_t j = blockIdx.x / s_count, slot = s_first + blockIdx.x % s_count, lane = threadIdx.x & 63, par = lane & 1;
    const M2DecTile d = info[j];
    if (!((d.kind == 1 && slot < 17) || (d.kind == 2 && slot == 17))) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const M2Blk mb = blk[(uint64_t)tile * M2_SLOTS + slot];
    uint8_t *out = scratch2 + sbase2[tile] + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot);
    const uint32_t type = sgpr(mb.type), n = sgpr(mb.n);
    const uint32_t Nnom = m2_nominal(slot);
    const int pb = slot >= 17 ? 15 : 14;
    if (type == 0 || n == 0) return;
    if (type == 1) {
        for (uint32_t i = lane; i < n; i += 64) out[i] = (uint8_t)mb.pbits;
        return;
    }
    if (type == 2) {  // raw symbols from b (libxpng.c:273)
        const uint32_t rb = (uint32_t)bit_width(Nnom - 1);
        const uint8_t *bw = d.blob + 8;
        const uint64_t endbit = (uint64_t)(d.bsz - 4) * 8;
        for (uint32_t i = lane; i < n; i += 64) out[i] = (uint8_t)bits_at(bw, (uint64_t)mb.pbits + (uint64_t)i * rb, rb, endbit);
        return;
    }
    if (type != 3 && type != 4) return;
    const uint8_t *blkp = d.blob + mb.cnt;
    const uint32_t size = sgpr(ld32u(blkp)) & 0xFFFFFF;
    if (size < 24) return;  // libxpng.c:285
    const uint8_t *words = blkp + 24;
    const uint32_t nw = (size - 24) >> 2;
    const uint16_t *F16 = tabs + ((uint64_t)tile * M2_SLOTS + slot) * 256;
    uint32_t hot0, hot1;
    {   // cum by 4-per-lane partial sums + wave scan; fc[i] = F | cum << 16
        const uint32_t b = lane * 4;
        const uint32_t f0 = b + 0 < Nnom ? F16[b + 0] : 0, f1 = b + 1 < Nnom ? F16[b + 1] : 0, f2 = b + 2 < Nnom ? F16[b + 2] : 0, f3 = b + 3 < Nnom ? F16[b + 3] : 0;
        const uint32_t tot = f0 + f1 + f2 + f3;
        uint32_t incl = tot;
        incl = wave_scan_incl(incl);
        const uint32_t c0 = incl - tot, c1 = c0 + f0, c2 = c1 + f1, c3 = c2 + f2;
        if (b + 0 < Nnom) fc[b + 0] = f0 | (c0 << 16);
        if (b + 1 < Nnom) fc[b + 1] = f1 | (c1 << 16);
        if (b + 2 < Nnom) fc[b + 2] = f2 | (c2 << 16);
        if (b + 3 < Nnom) fc[b + 3] = f3 | (c3 << 16);
        auto wmax = [&](uint32_t v) {
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { const uint32_t o = __shfl_xor(v, o2); v = o > v ? o : v; }
            return v;
        };
        const uint32_t k0 = (f0 << 8) | (b + 0), k1 = (f1 << 8) | (b + 1), k2 = (f2 << 8) | (b + 2), k3 = (f3 << 8) | (b + 3);
        uint32_t m = k0 > k1 ? k0 : k1; m = k2 > m ? k2 : m; m = k3 > m ? k3 : m;
        hot0 = wmax(m);
        auto ex = [&](uint32_t k) { return k == hot0 ? 0u : k; };
        uint32_t m2 = ex(k0) > ex(k1) ? ex(k0) : ex(k1); m2 = ex(k2) > m2 ? ex(k2) : m2; m2 = ex(k3) > m2 ? ex(k3) : m2;
        hot1 = wmax(m2);
    }
    __syncthreads();
    {   // coarse slot -> symbol: binary search over cum for every 8th slot (Nnom <= 256 -> 8 probes)
        const uint32_t groups = 1u << (pb - 3);
        for (uint32_t g = lane; g < groups; g += 64) {
            const uint32_t s = g << 3;
            uint32_t lo = 0, hi = Nnom - 1;  // largest index with cum <= s
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if ((fc[mid] >> 16) <= s) lo = mid; else hi = mid - 1;
            }
            while (lo > 0 && (fc[lo] & 0xFFFF) == 0) lo--;  // only reachable on corrupt tables
            coarse[g] = make_uint2(fc[lo], lo);
        }
    }
    __syncthreads();
    const uint32_t sym0 = hot0 & 255u, sym1 = hot1 & 255u;
    const uint32_t e0 = fc[sym0], e1 = fc[sym1];
    const uint32_t F0 = e0 & 0xFFFF, C0 = e0 >> 16, F1 = (hot1 >> 8) ? (e1 & 0xFFFF) : 0u, C1 = e1 >> 16;
    const uint32_t mask = (1u << pb) - 1;
    uint64_t s = ld64u(blkp + 8 + 8 * par);  // state0 at +8, state1 at +16
    uint32_t rw = 0;                          // scalar cursor: next word to read is words[rw]
    uint32_t ring_hi = nw < 512 ? nw : 512;   // ring holds word indices [ring_hi - 512, ring_hi) (those >= 0)
    for (uint32_t i = lane; i < ring_hi; i += 64) wring[i & 511u] = ld32u(words + 4ull * i);
    __syncthreads();
    rw = sgpr(rw); ring_hi = sgpr(ring_hi);
    // (step pieces as macros, not lambdas: see k_rans2_decode - closures used from several loops end up in scratch memory)
#define XPNG_D1_REFILL()                                                                                              \
    do {                                                                                                              \
        if (ring_hi < nw && rw + 128 > ring_hi) {                                                                     \
            const uint32_t new_hi_ = ring_hi + 256 < nw ? ring_hi + 256 : nw;                                         \
            __syncthreads();                                                                                          \
            for (uint32_t i_ = ring_hi + lane; i_ < new_hi_; i_ += 64) wring[i_ & 511u] = ld32u(words + 4ull * i_);   \
            ring_hi = new_hi_;                                                                                        \
            __syncthreads();                                                                                          \
        }                                                                                                             \
    } while (0)
#define XPNG_D1_FLUSH(base, hi)                                                                                       \
    do {                                                                                                              \
        __syncthreads();                                                                                              \
        for (uint32_t i_ = (base) + lane; i_ < (hi); i_ += 64) out[i_] = oring[i_ & 511u];                            \
        __syncthreads();                                                                                              \
    } while (0)
#define XPNG_D1_ONE(sym)                                                                                              \
    do {                                                                                                              \
        const uint32_t slt_ = (uint32_t)s & mask;                                                                     \
        const uint32_t d0_ = slt_ - C0, d1_ = slt_ - C1;                                                              \
        const bool hit0_ = d0_ < F0, hit1_ = d1_ < F1;                                                                \
        uint32_t F_, off_;                                                                                            \
        if (__ballot(!(hit0_ || hit1_)) == 0) { F_ = hit0_ ? F0 : F1; off_ = hit0_ ? d0_ : d1_; sym = hit0_ ? sym0 : sym1; } \
        else {                                                                                                        \
            const uint2 cg_ = coarse[slt_ >> 3];                                                                      \
            sym = cg_.y;                                                                                              \
            uint32_t e_ = cg_.x;                                                                                      \
            while (slt_ - (e_ >> 16) >= (e_ & 0xFFFF) && sym + 1 < Nnom) e_ = fc[++sym];                              \
            F_ = e_ & 0xFFFF; off_ = slt_ - (e_ >> 16);                                                               \
        }                                                                                                             \
        s = (uint64_t)F_ * (s >> pb) + off_;                                                                          \
    } while (0)
    // state0 refills first (libxpng.c:295-296); the cursor stops at the end of the block
#define XPNG_D1_RENORM(need, m)                                                                                       \
    do {                                                                                                              \
        const uint32_t n0_ = (m) & 1u, n1_ = (m) >> 1;                                                                \
        const uint32_t wsel_ = par ? (n0_ ? w2 : w1) : w1;                                                            \
        if (need) s = (s << 32) | wsel_;                                                                              \
        rw = rw + n0_ + n1_ < nw ? rw + n0_ + n1_ : nw;                                                               \
        XPNG_D1_REFILL();                                                                                             \
        w1 = wring[rw & 511u];                                                                                        \
        w2 = wring[(rw + 1) & 511u];                                                                                  \
    } while (0)
    const uint32_t pairs = sgpr(n >> 1);
    uint32_t w1 = wring[rw & 511u], w2 = wring[(rw + 1) & 511u];
    uint32_t flushed = 0, k = 0;
    // straight-line blocks of 4 pair steps (8 symbols): a lane packs its four symbols, the pair's 8 bytes are interleaved by a DPP
    // swap + two v_perm and leave as one 8-byte LDS store; the flush is checked once per block (2 k is a multiple of 8 here)
    for (; k + 4 <= pairs; k += 4) {
        uint32_t acc = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t sym = 0;
            XPNG_D1_ONE(sym);
            acc |= sym << (8 * u);
            const bool need = s < RANS_L;
            const uint32_t m = sgpr((uint32_t)__ballot(need) & 3u);
            if (m) XPNG_D1_RENORM(need, m);
        }
        const uint32_t oth = swap_pair(acc);
        const uint32_t lo = __builtin_amdgcn_perm(oth, acc, 0x05010400u), hi = __builtin_amdgcn_perm(oth, acc, 0x07030602u);
        if (lane == 0) *reinterpret_cast<uint2 *>(oring + ((2 * k) & 511u)) = make_uint2(lo, hi);
        if (((2 * k + 8) & 511u) == 0) { XPNG_D1_FLUSH(flushed, 2 * k + 8); flushed = 2 * k + 8; }
    }
    for (; k < pairs; k++) {
        uint32_t sym = 0;
        XPNG_D1_ONE(sym);
        const bool need = s < RANS_L;
        const uint32_t m = sgpr((uint32_t)__ballot(need) & 3u);
        if (lane < 2) oring[(2 * k + par) & 511u] = (uint8_t)sym;
        if (m) XPNG_D1_RENORM(need, m);
        if (((2 * k + 2) & 511u) == 0) { XPNG_D1_FLUSH(flushed, 2 * k + 2); flushed = 2 * k + 2; }
    }
    if (n & 1) {  // libxpng.c:300
        uint32_t sym = 0;
        XPNG_D1_ONE(sym);
        if (lane == 0) oring[(n - 1) & 511u] = (uint8_t)sym;
    }
    XPNG_D1_FLUSH(flushed, n);
#undef XPNG_D1_RENORM
#undef XPNG_D1_ONE
#undef XPNG_D1_FLUSH
#undef XPNG_D1_REFILL
}


// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_m2_dec_walk(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                    const uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                    const uint32_t *__restrict__ stream_n, uint8_t *__restrict__ nlseq) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x & 63;
    const M2DecTile d = info[j];
    if (d.kind != 1) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    // (the nine context streams start at multiples of 64 bytes inside the tile's scratch: the scalar-unit walk applies)
    ctx_walk_salu(scratch2 + sbase2[tile], lane < 9 ? (uint32_t)m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, lane) : 0, sgpr(d.coded), nlseq + t.pbase);
}

// --------------------------------------------------------------------------------------------------
// residual words per pixel.  colour: class streams by nl (DEC_, libxpng.c:901-910) + green add-back; gray: the single stream.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_m2_dec_resid(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                       const uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                       const uint32_t *__restrict__ stream_n, const uint8_t *__restrict__ nlseq,
                                                       uint32_t *__restrict__ resid) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const M2DecTile d = info[j];
    if (d.kind != 1 && d.kind != 2) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const uint8_t *sc = scratch2 + sbase2[tile];
    uint32_t *rs = resid + t.pbase;
    if (d.kind == 2) {
        const uint8_t *st = sc + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, 17);
        for (uint32_t i = tid; i < t.n; i += THREADS) {
            uint32_t w = 0;
            if (i) { const uint32_t v = (uint32_t)zz_dec(st[i - 1]) & 255u; w = v | (v << 8) | (v << 16) | (1u << 24); }
            rs[i] = w;
        }
        return;
    }
    // colour: a lane owns FOUR consecutive pixels (their nl symbols are four consecutive bytes of the nl sequence: every pixel but the
    // first is coded); the position of a pixel's symbol(s) in its class stream is a prefix sum of eight counters packed three 10-bit
    // fields to a word: three DPP scans per 4 x THREADS pixels (rounds 1-3: one pixel per lane, eight ballots per THREADS pixels);
    // the four residual words leave as one 16-byte store.  Two barriers per iteration.
    const uint32_t *nls4 = reinterpret_cast<const uint32_t *>(nlseq + t.pbase);  // (plane bases are multiples of 256)
    const int useG = d.m & 1;
    constexpr int WAVES = THREADS / 64, PX = THREADS * 4;
    __shared__ uint32_t s_wave[WAVES][9], s_run[9], s_off[9];
    __shared__ volatile uint32_t s_base[WAVES][16];
    if (tid < 9) s_run[tid] = 0;
    if (tid >= 1 && tid < 9) s_off[tid] = (uint32_t)m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, 8 + tid);  // class stream of nl = tid
    __syncthreads();
    const uint32_t myq = __umul24(lane, 11u) >> 5, mysh = __umul24(lane - __umul24(myq, 3u), 10u);  // field of counter `lane` in the packed words
    uint32_t py = (4 * tid) / t.w, px = 4 * tid - py * t.w;   // (x, y) of the lane's first pixel, advanced by PX pixels per iteration
    const uint32_t dy = PX / t.w, dx = PX - dy * t.w;
    for (uint32_t i0 = 0; i0 < t.n; i0 += PX) {
        const uint32_t i = i0 + 4 * tid;
        // nl of pixels i .. i+3 = bytes i-1 .. i+2 of the nl sequence (pixel 0 has none)
        uint32_t u = 0;
        if (i < t.n) {
            const uint32_t hi = nls4[i >> 2], lo = i ? nls4[(i >> 2) - 1] : 0u;
            u = __builtin_amdgcn_alignbyte(hi, lo, 3);
        }
        uint32_t nl[4], csh[4], cinc[4][3], c0 = 0, c1 = 0, c2 = 0;
        bool coded[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            coded[k] = i + k < t.n && i + k > 0;
            nl[k] = coded[k] ? min((u >> (8 * k)) & 255u, 8u) : 0u;
            const uint32_t v = nl[k], q = __umul24(v, 11u) >> 5;
            csh[k] = __umul24(v - __umul24(q, 3u), 10u);
            const uint32_t one = v ? 1u << csh[k] : 0u;
            cinc[k][0] = q == 0 ? one : 0u; cinc[k][1] = q == 1 ? one : 0u; cinc[k][2] = q == 2 ? one : 0u;
            c0 += cinc[k][0]; c1 += cinc[k][1]; c2 += cinc[k][2];
        }
        const uint32_t y0 = wave_scan_incl(c0), y1 = wave_scan_incl(c1), y2 = wave_scan_incl(c2);
        {
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)y0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)y1, 63),
                           t2 = (uint32_t)__builtin_amdgcn_readlane((int)y2, 63);
            const uint32_t word = myq == 0 ? t0 : (myq == 1 ? t1 : t2);
            if (lane < 9) s_wave[wv][lane] = (word >> mysh) & 1023u;
        }
        __syncthreads();  // (A) per-wave class counts visible
        if (lane < 9) {
            uint32_t bk = s_run[lane];
            for (uint32_t w = 0; w < wv; w++) bk += s_wave[w][lane];
            s_base[wv][lane] = bk;
        }
        __builtin_amdgcn_wave_barrier();
        if (i < t.n) {
            uint32_t f0 = y0 - c0, f1 = y1 - c1, f2 = y2 - c2, word[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t w = 0;
                if (coded[k]) {
                    int dr = 0, dg = 0, db = 0;
                    const uint32_t v = nl[k];
                    if (v) {
                        const uint32_t f = cinc[k][0] ? f0 : (cinc[k][1] ? f1 : f2);
                        const uint32_t kk = s_base[wv][v] + ((f >> csh[k]) & 1023u);
                        const uint8_t *st = sc + s_off[v];
                        uint32_t zr, zg, zb;
                        if (v == 1) { const uint32_t b = st[kk]; zr = b >> 2; zg = (b >> 1) & 1; zb = b & 1; }
                        else if (v == 2) { const uint32_t b = st[kk]; zr = b >> 4; zg = (b >> 2) & 3; zb = b & 3; }
                        else { zr = st[3 * kk]; zg = st[3 * kk + 1]; zb = st[3 * kk + 2]; }
                        dr = zz_dec((int)zr); dg = zz_dec((int)zg); db = zz_dec((int)zb);
                    }
                    uint32_t x = px + k, y = py;
                    if (x >= t.w) { x -= t.w; y++; }   // (tiles at least 4 wide: one wrap at most; narrower tiles below)
                    if (t.w < 4) { y = (i + k) / t.w; x = (i + k) - y * t.w; }
                    if (useG && x > 0 && y > 0) { dr += dg; db += dg; }
                    w = ((uint32_t)dr & 255u) | (((uint32_t)dg & 255u) << 8) | (((uint32_t)db & 255u) << 16) | (1u << 24);
                }
                word[k] = w;
                f0 += cinc[k][0]; f1 += cinc[k][1]; f2 += cinc[k][2];
            }
            *reinterpret_cast<u32x4_t *>(rs + i) = u32x4_t{word[0], word[1], word[2], word[3]};  // (the residual plane carries slack behind a tile)
        }
        uint32_t tot = 0;
        if (tid >= 1 && tid < 9) for (int w2 = 0; w2 < WAVES; w2++) tot += s_wave[w2][tid];
        __syncthreads();  // (C) every wave has read the running and the per-wave counts
        if (tid >= 1 && tid < 9) s_run[tid] += tot;
        px += dx; py += dy;
        if (px >= t.w) { px -= t.w; py++; }
        // (the next iteration's barrier (A) orders these LDS writes before their next use)
    }
}

// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_m2_dec_recon(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                       const uint32_t *__restrict__ resid, uint8_t *const *__restrict__ rasters,
                                                       uint64_t bpr, uint32_t free_ew) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const M2DecTile d = info[j];
    const TileDesc t = tiles[vtile(sel, j)];
    uint8_t *dst = rasters[t.img] + (uint64_t)t.y * bpr + (uint64_t)t.x * 3;
    const uint64_t row = (uint64_t)t.w * 3;
    if (d.kind == M2_KIND_BAD) return;
    if (d.kind == 0) {  // raw rows (libxpng.c:941)
        const uint8_t *src = d.blob + 4;
        for (uint64_t b = tid; b < row * t.h; b += blockDim.x) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = src[b]; }
        return;
    }
    if (d.kind == 4) {  // single colour (libxpng.c:916-927)
        const uint8_t *px = d.blob + 4;
        for (uint64_t b = tid; b < row * t.h; b += blockDim.x) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = px[o % 3]; }
        return;
    }
    if (d.kind == 3) {  // raw gray (libxpng.c:875-878)
        const uint8_t *src = d.blob + 4;
        for (uint64_t b = tid; b < row * t.h; b += blockDim.x) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = src[y * t.w + o / 3]; }
        return;
    }
    const uint32_t w0 = ld32u(d.blob + 8);  // head of b: first pixel, MSB first
    uint32_t first;
    int predmode;
    if (d.kind == 2) {
        const uint32_t g = w0 >> 24;
        first = g | (g << 8) | (g << 16);
        predmode = d.m == 0 ? 2 : d.m == 1 ? 3 : d.m == 2 ? 0 : 1;  // p1x, p1y, p2a, p3a (libxpng.c:890-895)
    } else {
        first = ((w0 >> 24) & 255u) | (((w0 >> 16) & 255u) << 8) | (((w0 >> 8) & 255u) << 16);
        predmode = (d.m >> 1) & 1;
    }
    if (free_ew) {
        extern __shared__ uint32_t dyn_lds[];
        recon_free<3>(t, dst, bpr, nullptr, resid + t.pbase, first, predmode, dyn_lds + 16, free_ew, dyn_lds);
    } else {
        __shared__ uint32_t s_row[2][1024];
        recon_wavefront<3>(t, dst, bpr, nullptr, resid + t.pbase, first, predmode, s_row);
    }
}

// The same for batches: ONE wavefront per tile sweeping 64-row bands (recon_band_core, m1_decode.hpp: lane = row, U by DPP, byte-
// parallel predictors, 16-byte residual loads and 12-byte pixel stores), as mode 1 uses; the multi-wave form above hands rows
// from wave to wave through LDS and progress counters, which is right for one image and 4x slower per tile in a batch.
__global__ __launch_bounds__(64) void k_m2_dec_recon_band(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                          const uint32_t *__restrict__ resid, uint8_t *const *__restrict__ rasters, uint64_t bpr) {
    extern __shared__ uint32_t rb_lds[];  // staging rings, then the seam row (recon_band_core)
    uint32_t *seam = rb_lds + RB_STAGE_WORDS;
    const uint32_t j = blockIdx.x, lane = threadIdx.x & 63;
    const M2DecTile *d = info + j;
    const uint32_t kind = d->kind;
    if (kind == M2_KIND_BAD) return;
    const TileDesc t = tiles[vtile(sel, j)];
    uint8_t *dst = rasters[t.img] + (uint64_t)t.y * bpr + (uint64_t)t.x * 3;
    const uint64_t row = (uint64_t)t.w * 3;
    if (kind == 0 || kind == 3 || kind == 4) {  // raw rows (libxpng.c:941), raw gray (875-878), single colour (916-927)
        const uint8_t *src = d->blob + 4;
        for (uint64_t b = lane; b < row * t.h; b += 64) {
            const uint64_t y = b / row, o = b - y * row;
            dst[y * bpr + o] = kind == 0 ? src[b] : kind == 3 ? src[y * t.w + o / 3] : src[o % 3];
        }
        return;
    }
    const uint32_t w0 = ld32u(d->blob + 8);  // head of b: first pixel, MSB first
    uint32_t first;
    int predmode;
    if (kind == 2) {
        const uint32_t g = w0 >> 24;
        first = g | (g << 8) | (g << 16);
        predmode = d->m == 0 ? 2 : d->m == 1 ? 3 : d->m == 2 ? 0 : 1;  // p1x, p1y, p2a, p3a (libxpng.c:890-895)
    } else {
        first = ((w0 >> 24) & 255u) | (((w0 >> 16) & 255u) << 8) | (((w0 >> 8) & 255u) << 16);
        predmode = (int)((d->m >> 1) & 1);
    }
    recon_band_core<3>(t, dst, bpr, resid + t.pbase, first, predmode, rb_lds, seam, 0u);
}

inline int m2_wide_decode(DecodeWs &ws, uint32_t B, uint64_t n_tiles, uint32_t total, const M2DecTile *d_info2, const TileDesc *d_tiles,
                          TileSel sel, const M2Blk *d_blk2, const uint16_t *d_tabs2, uint8_t *d_scratch2, const uint64_t *d_sbase2,
                          const uint32_t *d_stream_n2, hipStream_t s, std::string &err);  // rans1_wide_dec.hpp

// Launch the mode-2 decode of tiles [t0, t1) of every image of the batch (RGB only).
inline int decode_m2_launch(DecodeWs &ws, uint32_t B, uint64_t n_tiles, uint64_t plane_total, const TileDesc *d_tiles, uint64_t W,
                            uint32_t max_w, uint32_t max_h,
                            const uint8_t *const *d_blob_ptrs, const uint64_t *d_blob_len, uint32_t *d_status,
                            const uint64_t *tile_off, uint32_t t0, uint32_t t1,
                            uint8_t *const *d_raster_ptrs, M2DecTile *d_info2, M2Blk *d_blk2, uint16_t *d_tabs2, uint8_t *d_scratch2,
                            const uint64_t *d_sbase2, uint32_t *d_stream_n2, hipStream_t s, std::string &err) {
    const uint32_t cnt = t1 - t0, total = B * cnt;
    const TileSel sel{t0, cnt, (uint32_t)n_tiles, B, nullptr};
    if (decode_ws_prepare(ws, B, n_tiles, plane_total, tile_off, t0, total, s, err, d_blob_ptrs, d_blob_len)) return 1;
    if (!ws.d_off || !ws.d_nlseq || !ws.d_resid || !d_info2 || !d_blk2 || !d_tabs2 || !d_scratch2 || !d_sbase2 || !d_stream_n2 || !d_tiles || !d_blob_ptrs || !d_raster_ptrs) {
        err = "internal error: a mode-2 decode workspace buffer was never allocated";  // (a null pointer in a kernel is a GPU fault = abort())
        return 1;
    }
    const uint64_t bpr = W * 3;
    if (hipMemsetAsync(d_blk2, 0, (uint64_t)B * n_tiles * M2_SLOTS * sizeof(M2Blk), s) != hipSuccess) { err = "memset failed"; return 1; }
    k_m2_dec_parse<<<(total + 63) / 64, 64, 0, s>>>(d_blob_ptrs, ws.d_off, d_blob_len, cnt, total, d_tiles, sel, d_info2, d_blk2, d_tabs2, d_stream_n2, d_status);
    if (getenv("XPNG_NARROW_RANS") || ((uint64_t)total * M2_STREAMS <= 2048 && !getenv("XPNG_WIDE_RANS"))) {
        k_rans1_decode<14><<<total * M2_STREAMS, 64, 0, s>>>(d_info2, d_tiles, sel, 0, M2_STREAMS, d_blk2, d_tabs2, d_scratch2, d_sbase2, d_stream_n2);
        k_rans1_decode<15><<<total, 64, 0, s>>>(d_info2, d_tiles, sel, 17, 1, d_blk2, d_tabs2, d_scratch2, d_sbase2, d_stream_n2);  // gray tiles
    } else if (m2_wide_decode(ws, B, n_tiles, total, d_info2, d_tiles, sel, d_blk2, d_tabs2, d_scratch2, d_sbase2, d_stream_n2, s, err)) return 1;
    k_m2_dec_walk<<<total, 64, 0, s>>>(d_info2, d_tiles, sel, d_scratch2, d_sbase2, d_stream_n2, ws.d_nlseq);
    if ((uint64_t)total * M2_STREAMS > 2048 && !probe_env("XPNG_BIG_BLOCKS")) k_m2_dec_resid<256><<<total, 256, 0, s>>>(d_info2, d_tiles, sel, d_scratch2, d_sbase2, d_stream_n2, ws.d_nlseq, ws.d_resid);
    else k_m2_dec_resid<1024><<<total, 1024, 0, s>>>(d_info2, d_tiles, sel, d_scratch2, d_sbase2, d_stream_n2, ws.d_nlseq, ws.d_resid);
    uint32_t free_ew, rthreads, rlds;
    recon_geometry(max_w, max_h, free_ew, rthreads, rlds);
    const bool wide_recon = !getenv("XPNG_NARROW_RANS") && ((uint64_t)total * M2_STREAMS > 2048 || getenv("XPNG_WIDE_RANS")) && max_w <= RB_MAXW && !probe_env("XPNG_WAVEFRONT_RECON");
    if (wide_recon) k_m2_dec_recon_band<<<total, 64, RB_LDS_BYTES(max_w), s>>>(d_info2, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr);
    else k_m2_dec_recon<<<total, rthreads, rlds, s>>>(d_info2, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr, free_ew);
    if (hipGetLastError() != hipSuccess) { err = "mode-2 decode kernel launch failed"; return 1; }
    return 0;
}

}  // namespace xpng
