// m1_decode.hpp -- mode-1 tile DECODE kernels for gfx950.
//
// Reference path restated: dec_1_th (libxpng.c:834-863) = decompress_block_v2 (429-493) + m1d_* (796-830).
// The reference decodes a tile in one serial loop; its three serial couplings are separated here:
//
//   k_dec_parse       tile header, k extent, the 9(+1) v2 block offsets / symbol counts     (1 thread / tile)
//   k_rans2_decode    one wavefront per (tile, stream); 2 lanes carry the interleaved states (429-493)
//   k_dec_alpha       alpha plane = column-0 prefix sum, then per-row prefix sums mod 256   (798-800: alpha uses
//                     the left neighbour everywhere except column 0)
//   k_dec_walk        the context chain nl = *cx[nl]++ (803): strictly serial per tile; lanes 0..8 of one wave
//                     own the nine queues in registers and the chain advances by v_readlane
//   k_dec_resid       bit cursor = scan of 3*nl; residual extraction from k; zig-zag / green add-back (804-813)
//   k_dec_recon       causal prediction from reconstructed L/U/UL: anti-diagonal wavefront, one row per thread
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace xpng {

struct DecTile {  // per-tile parse result (device)
    const uint8_t *blob;  // the tile blob (inside its image's blob buffer)
    uint32_t type;      // tile type byte (0 = raw rows)
    uint32_t kbytes;    // bytes of k words
    uint32_t blk_off[10];  // block offsets relative to the blob start
    uint32_t blk_n[10];    // symbols per block
    uint32_t ctx_start[10];  // 16-byte aligned start of context stream c inside the tile's symbol area (ctx_start[9] = total coded)
};

struct WDec;  // rans2_wide_dec.hpp

struct DecodeWs {
    uint64_t cap_tiles = 0, cap_plane = 0;
    WDec *d_wdec = nullptr;              // wide rANS decode: per (tile, stream) descriptors and decode tables
    uint8_t *d_dtab = nullptr;
    WDec *d_wdec2 = nullptr;             // the same for mode 2 (21 slots per tile), allocated on first use
    uint8_t *d_dtab2 = nullptr;
    uint64_t cap2 = 0;
    std::vector<uint64_t> last_off;      // tile offsets already resident in d_off (skip the upload when unchanged)
    uint32_t last_t0 = 0;
    hipStream_t side = nullptr;          // alpha branch runs beside the nl-context branch
    bool side_borrowed = false;          // `side` is the context's one side stream (the encode's alpha chains use it too, never at the same time): not ours to destroy
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t side2 = nullptr;         // walk / residuals / reconstruction of the smaller tiles, beside the walk of the biggest
    hipEvent_t ev_ctx = nullptr, ev_small = nullptr;
    DecTile *d_info = nullptr;
    uint64_t *d_off = nullptr;
    uint8_t *d_ctxsym = nullptr, *d_asym = nullptr, *d_alpha = nullptr, *d_nlseq = nullptr;
    uint32_t *d_resid = nullptr, *d_resid_alloc = nullptr;  // d_resid = d_resid_alloc + 16: k_dec_recon_band reads up to 3 words before a tile's first
    // The five symbol / residual planes (8 B per pixel) are carved out of `arena` when it is big enough: the context hands
    // in its encode stream scratch, which no decode kernel reads and no encode kernel touches while a decode of the same
    // context runs (calls on one context are ordered on one stream).  planes_in_arena: those pointers are not ours to free.
    uint8_t *arena = nullptr;
    uint64_t arena_bytes = 0;
    bool planes_in_arena = false;
};
inline void decode_ws_free(DecodeWs &w) {
    void *p[] = {w.d_info, w.d_off, w.d_wdec, w.d_dtab, w.d_wdec2, w.d_dtab2};
    for (void *q : p) if (q) (void)hipFree(q);
    void *planes[] = {w.d_ctxsym, w.d_asym, w.d_alpha, w.d_nlseq, w.d_resid_alloc};
    for (void *q : planes) if (q && !w.planes_in_arena) (void)hipFree(q);
    if (w.side && !w.side_borrowed) (void)hipStreamDestroy(w.side);
    if (w.side2) (void)hipStreamDestroy(w.side2);
    if (w.ev_ctx) (void)hipEventDestroy(w.ev_ctx);
    if (w.ev_small) (void)hipEventDestroy(w.ev_small);
    if (w.ev_fork) (void)hipEventDestroy(w.ev_fork);
    if (w.ev_join) (void)hipEventDestroy(w.ev_join);
    uint8_t *arena = w.arena;
    const uint64_t arena_bytes = w.arena_bytes;
    hipStream_t lent = w.side_borrowed ? w.side : nullptr;
    w = DecodeWs();  // (also clears last_off)
    w.arena = arena; w.arena_bytes = arena_bytes;
    if (lent) { w.side = lent; w.side_borrowed = true; }  // (what the context lent stays lent)
}

// unaligned-safe little-endian u32 load from global memory (tile blobs are only byte-aligned after a raw RGB tile)
__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3) * 8;
    const uint32_t lo = q[0];
    if (sh == 0) return lo;
    return (lo >> sh) | (q[1] << (32 - sh));
}
__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { return (uint64_t)ld32u(p) | ((uint64_t)ld32u(p + 4) << 32); }

// MSB-first bit reader over u32 words (BITSTREAM_FILL / BITSTREAM_READ, libxpng.c:9,12); words past `end` read as 0
struct BitR {
    uint64_t acc;
    uint32_t have;
    const uint8_t *p, *end;
    __device__ __forceinline__ uint32_t get(uint32_t c) {
        if (have < 32) {
            acc <<= 32; have += 32;
            if (p < end) { acc += ld32u(p); p += 4; }
        }
        have -= c;
        return (uint32_t)((acc >> have) & ((1ull << c) - 1));
    }
};

// --------------------------------------------------------------------------------------------------
constexpr uint32_t TILE_BAD = 0xEE;  // type value of a tile whose headers failed validation: every kernel skips it

// The reference decoder trusts the file ("no bounds checks on file contents"); on a GPU a wild
// offset is a fault that can take the device down, so every length / offset / count is checked here, once, and a bad
// tile is skipped (its pixels stay untouched) with bit 0 of *status set.
__global__ void k_dec_parse(const uint8_t *const *__restrict__ blobs, const uint64_t *__restrict__ off,
                            const uint64_t *__restrict__ blob_len, uint32_t cnt, uint32_t total, uint32_t spt, int pxsz,
                            const TileDesc *__restrict__ tiles, TileSel sel, DecTile *__restrict__ info,
                            uint32_t *__restrict__ status) {
    bw_prio();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t vt = vtile(sel, j), il = imglin(sel, vt);  // off[] is image-major (host order)
    const TileDesc t = tiles[vt];
    DecTile d{};
    d.blob = blobs[t.img] + off[il];
    const uint64_t avail = blob_len[t.img] > off[il] ? blob_len[t.img] - off[il] : 0;
    const uint8_t *f = d.blob;
    bool ok = avail >= 4;
    uint32_t L = 0;
    if (ok) {
        const uint32_t h0 = ld32u(f);
        d.type = h0 >> 24;
        L = h0 & 0xFFFFFF;
        ok = L <= avail;
    }
    if (ok && d.type == 0) ok = L == t.n * (uint32_t)pxsz + 4;
    else if (ok) {
        ok = L >= 12 && (d.type >> 4) == 1;
        const uint32_t ksz = ok ? ld32u(f + 4) : 0;  // includes itself (libxpng.c:556)
        ok = ok && ksz >= 8 && (ksz & 3) == 0 && 4 + (uint64_t)ksz <= L;
        d.kbytes = ksz - 4;
        uint64_t o = 4 + (uint64_t)ksz;
        uint32_t acc = 0, coded = 0;  // acc: 16-byte aligned start of the next context stream inside the tile's symbol area
        for (uint32_t c = 0; c < spt && ok; c++) {
            ok = o + 4 <= L;
            if (!ok) break;
            const uint32_t b0 = ld32u(f + o), ty = b0 >> 24, sz = ty == 0 ? 4 : (b0 & 0xFFFFFF);
            ok = ty <= 4 && sz >= (ty == 0 ? 4u : 8u) && (sz & 3) == 0 && o + sz <= L;
            if (!ok) break;
            const uint32_t w1 = ty ? ld32u(f + o + 4) : 0, n = w1 & 0xFFFFFF, v2 = w1 >> 24;
            if (ty == 2) ok = v2 >= 1 && v2 <= 8 && 8 + 4 * (((uint64_t)n * v2 + 31) >> 5) <= sz;
            if (ty >= 3) {
                ok = sz >= 28 + 4 && v2 <= 254;
                if (ok) {
                    const uint32_t h2 = ld32u(f + o + 8), toff = h2 & 0xFFFFFF, pb = h2 >> 24;
                    ok = pb >= 10 && pb <= 15 && toff >= 5 && 8 + 4 * (uint64_t)toff < sz;
                }
            }
            d.blk_off[c] = (uint32_t)o;
            d.blk_n[c] = ty == 0 ? 0 : n;
            if (c < 9) { d.ctx_start[c] = acc; acc = (acc + d.blk_n[c] + 15u) & ~15u; coded += d.blk_n[c]; ok = ok && coded <= t.n - 1; }
            else ok = ok && d.blk_n[c] <= t.n - 1;
            o += sz;
        }
        d.ctx_start[9] = coded;
    }
    if (!ok) { d.type = TILE_BAD; atomicOr(status, 1u); }
    info[j] = d;
}

}  // namespace xpng
#include "rans2_wide_dec.hpp"
namespace xpng {

// --------------------------------------------------------------------------------------------------
// one v2 block -> symbols (decompress_block_v2, libxpng.c:429-493).  Single-wave workgroup per (tile, stream).
//
// The recurrence runs backwards over the symbols with two interleaved states; even lanes carry state0, odd lanes
// state1 (lanes 2..63 replicate lanes 0/1, so nothing in the loop needs a lane mask except the stores).  The
// dependent chain per step is kept short:
//   * hot-symbol cache: the two most probable symbols' (cum, F) live in registers; a step whose slot falls into one of
//     them needs no table lookup at all (alpha and nl streams are heavily skewed).  Otherwise slot -> symbol comes
//     from an LDS byte table and (F, cum) from a 256-entry LDS table;
//   * renormalisation words: the cursor is a SCALAR (it moves by popcount of a 2-bit ballot; state1 pops first as in
//     libxpng.c:486-487); words are staged coalesced into an LDS ring and the two candidates words[rw-1], words[rw-2]
//     are read speculatively at the top of the step, off the dependent chain;
//   * decoded symbols go to an LDS ring and are flushed 512 at a time.
// MAXPB bounds the slot table (2^MAXPB bytes of LDS): 12 for the nl-context streams, 15 for alpha.  A block whose
// header asks for more than MAXPB is left to the general (MAXPB = 15) launch (`only_over` selects those).
// One rANS v2 block, one wavefront (the body of k_rans2_decode and of k_rans2_decode_rest below).
// COARSE = false: no slot table at all - a slot is resolved by a binary search over the cumulative counts.  Slow, but 4.6 KB of
// LDS instead of 37: the form k_rans2_decode_rest uses (streams the reference never writes; its launch sits on the critical path of
// every batched decode and used to wait milliseconds for 128 x 37 KB of LDS to find nothing to do).
template <int MAXPB, bool COARSE = true>
__device__ __forceinline__ void rans2_decode_stream(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                    const uint32_t j, const uint32_t c, int only_over, uint8_t *__restrict__ ctxsym,
                                                    uint8_t *__restrict__ asym, uint64_t *__restrict__ dbg) {
    // per group of 8 slots: (F | cum << 16, symbol) of the symbol owning slot (g << 3): ONE LDS read resolves a slot whose group
    // lies inside one symbol's range (wide symbols: the probable ones), else a short forward scan over fc[] follows
    __shared__ uint2 coarse[COARSE ? 1 << (MAXPB - 3) : 1];
    __shared__ uint32_t fc[256];
    __shared__ uint32_t Fs[260];
    __shared__ uint32_t wring[512];
    __shared__ __align__(8) uint8_t oring[512];
    const uint32_t lane = threadIdx.x & 63, par = lane & 1;
    const DecTile d = info[j];
    if (d.type == 0 || d.type == TILE_BAD) return;
    const uint32_t vt = vtile(sel, j);
    const TileDesc t = tiles[vt];
    const uint8_t *in = d.blob + d.blk_off[c];
    uint8_t *out = c < 9 ? ctxsym + t.pbase + d.ctx_start[c] : asym + t.pbase;
    const uint32_t h0 = sgpr(ld32u(in)), type = h0 >> 24;  // header words are wave-uniform: keep them (and every loop
    if (type == 0) return;                                  // cursor derived from them) in SGPRs
    const uint32_t csz = h0 & 0xFFFFFF;
    const uint8_t *end = in + csz;
    const uint32_t h1 = sgpr(ld32u(in + 4)), n = h1 & 0xFFFFFF, v2 = h1 >> 24;
    if (type <= 2 && only_over) return;
    if (type == 1) {  // one distinct symbol
        for (uint32_t i = lane; i < n; i += 64) out[i] = (uint8_t)v2;
        return;
    }
    if (type == 2) {  // raw: v2 bits per symbol, MSB first
        for (uint32_t i = lane; i < n; i += 64) {
            const uint64_t b0 = (uint64_t)i * v2;
            const uint8_t *wp = in + 8 + (b0 >> 5) * 4;
            const uint32_t rel = (uint32_t)(b0 & 31);
            const uint64_t two = ((uint64_t)ld32u(wp) << 32) | (wp + 4 < end ? ld32u(wp + 4) : 0u);
            out[i] = (uint8_t)((two >> (64 - rel - v2)) & ((1u << v2) - 1));
        }
        return;
    }
    const uint32_t N = v2 + 2;
    const uint32_t h2 = sgpr(ld32u(in + 8));
    const int pb = (int)(h2 >> 24);
    if ((pb > MAXPB) || (only_over == 1 && pb <= 12)) return;  // handled by the other launch
    uint64_t *stamp = dbg ? dbg + ((uint64_t)vt * 10 + c) * 8 : nullptr;
#define XPNG_DSTAMP(k) do { if (stamp && lane == 0) stamp[k] = __builtin_readcyclecounter(); } while (0)
    XPNG_DSTAMP(0);
    const uint8_t *words = in + 12;
    const uint8_t *table = in + 8 + 4ull * (h2 & 0xFFFFFF);
    if (lane == 0) {  // frequency table: <= 256 short fields, serial bit reader
        BitR tr{0, 0, table, end};
        for (uint32_t i = 0; i < N; i++) {
            uint32_t F;
            if (type == 3) F = tr.get((uint32_t)pb);
            else F = tr.get(1) ? tr.get((uint32_t)pb) : 0;
            Fs[i] = F;
        }
    }
    __syncthreads();
    uint32_t hot0, hot1;  // (F << 8 | sym) of the two most probable symbols
    {   // cum by 4-per-lane partial sums + wave scan; fc[i] = F | cum << 16
        const uint32_t b = lane * 4;
        const uint32_t f0 = b + 0 < N ? Fs[b + 0] : 0, f1 = b + 1 < N ? Fs[b + 1] : 0, f2 = b + 2 < N ? Fs[b + 2] : 0, f3 = b + 3 < N ? Fs[b + 3] : 0;
        const uint32_t tot = f0 + f1 + f2 + f3;
        uint32_t incl = tot;
        incl = wave_scan_incl(incl);
        const uint32_t c0 = incl - tot, c1 = c0 + f0, c2 = c1 + f1, c3 = c2 + f2;
        if (b + 0 < N) fc[b + 0] = f0 | (c0 << 16);
        if (b + 1 < N) fc[b + 1] = f1 | (c1 << 16);
        if (b + 2 < N) fc[b + 2] = f2 | (c2 << 16);
        if (b + 3 < N) fc[b + 3] = f3 | (c3 << 16);
        auto wmax = [&](uint32_t v) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(v, off); v = o > v ? o : v; }
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
    if (COARSE) {   // coarse slot -> symbol: binary search over cum for every 8th slot (N <= 256 -> 8 probes)
        const uint32_t groups = 1u << (pb - 3);
        for (uint32_t g = lane; g < groups; g += 64) {
            const uint32_t s = g << 3;
            uint32_t lo = 0, hi = N - 1;  // largest index with cum <= s
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if ((fc[mid] >> 16) <= s) lo = mid; else hi = mid - 1;
            }
            while (lo > 0 && (fc[lo] & 0xFFFF) == 0) lo--;  // only reachable on corrupt tables
            coarse[g] = make_uint2(fc[lo], lo);
        }
    }
    __syncthreads();
    XPNG_DSTAMP(1);
    const uint32_t sym0 = hot0 & 255u, sym1 = hot1 & 255u;
    const uint32_t e0 = fc[sym0], e1 = fc[sym1];
    const uint32_t F0 = e0 & 0xFFFF, C0 = e0 >> 16, F1 = (hot1 >> 8) ? (e1 & 0xFFFF) : 0u, C1 = e1 >> 16;
    const uint32_t mask = (1u << pb) - 1;
    const uint8_t *sp = table - 16;  // state0 at table-16, state1 at table-8
    const uint32_t nw = (uint32_t)((sp - words) >> 2);  // renormalisation words below the states
    uint32_t rw = nw;                                    // scalar cursor: next word to pop is words[rw-1]
    uint32_t ring_lo = nw > 512 ? nw - 512 : 0;          // ring holds word indices [ring_lo, ring_lo + 512)
    for (uint32_t i = ring_lo + lane; i < nw; i += 64) wring[i & 511u] = ld32u(words + 4ull * i);
    uint64_t s = ld64u(sp + 8 * par);
    __syncthreads();
    // The step pieces below are MACROS, not lambdas: with by-reference closures used from three loops (prologue, straight-line
    // blocks, epilogue) the compiler kept the closures - and with them the state, the cursors and the word candidates - in
    // scratch memory behind pointers (ScratchSize 464, a scratch access every few instructions of the chain).
    // uniform: bring the next lower 256 words into the ring before the cursor reaches them
#define XPNG_DEC_REFILL()                                                                                         \
    do {                                                                                                          \
        if (ring_lo > 0 && rw < ring_lo + 128) {                                                                  \
            const uint32_t new_lo_ = ring_lo > 256 ? ring_lo - 256 : 0;                                           \
            __syncthreads();                                                                                      \
            for (uint32_t i_ = new_lo_ + lane; i_ < ring_lo; i_ += 64) wring[i_ & 511u] = ld32u(words + 4ull * i_); \
            ring_lo = new_lo_;                                                                                    \
            __syncthreads();                                                                                      \
        }                                                                                                         \
    } while (0)
    // uniform: symbols [base, min(base+512, n)) leave the LDS ring
#define XPNG_DEC_FLUSH(base)                                                                                      \
    do {                                                                                                          \
        __syncthreads();                                                                                          \
        const uint32_t hi_ = (base) + 512 < n ? (base) + 512 : n;                                                 \
        for (uint32_t i_ = (base) + lane; i_ < hi_; i_ += 64) out[i_] = oring[i_ & 511u];                          \
        __syncthreads();                                                                                          \
    } while (0)
    // one symbol out of this lane's state -> sym; need = the state must refill
#define XPNG_DEC_ONE(sym, need)                                                                                   \
    do {                                                                                                          \
        const uint32_t slot_ = (uint32_t)s & mask;                                                                \
        const uint32_t d0_ = slot_ - C0, d1_ = slot_ - C1;                                                        \
        const bool hit0_ = d0_ < F0, hit1_ = d1_ < F1;                                                            \
        uint32_t F_, off_;                                                                                        \
        if (__ballot(!(hit0_ || hit1_)) == 0) {                                                                   \
            F_ = hit0_ ? F0 : F1; off_ = hit0_ ? d0_ : d1_; sym = hit0_ ? sym0 : sym1;                            \
        } else {                                                                                                  \
            uint32_t e_;                                                                                          \
            if (COARSE) { const uint2 cg_ = coarse[slot_ >> 3]; sym = cg_.y; e_ = cg_.x; }                        \
            else {  /* largest index with cum <= slot */                                                          \
                uint32_t lo_ = 0, hi_ = N - 1;                                                                    \
                while (lo_ < hi_) { const uint32_t mid_ = (lo_ + hi_ + 1) >> 1; if ((fc[mid_] >> 16) <= slot_) lo_ = mid_; else hi_ = mid_ - 1; } \
                while (lo_ > 0 && (fc[lo_] & 0xFFFF) == 0) lo_--;                                                 \
                sym = lo_; e_ = fc[lo_];                                                                          \
            }                                                                                                     \
            while (slot_ - (e_ >> 16) >= (e_ & 0xFFFF) && sym + 1 < N) e_ = fc[++sym];  /* to the symbol whose [cum, cum+F) holds the slot */ \
            F_ = e_ & 0xFFFF; off_ = slot_ - (e_ >> 16);                                                          \
        }                                                                                                         \
        s = (uint64_t)F_ * (s >> pb) + off_;                                                                      \
        need = s < RANS_L;                                                                                        \
    } while (0)
    // uniform, rare: state1 refills first (libxpng.c:486-487); the candidates words[rw-1], words[rw-2] live in registers and are
    // re-read from the LDS ring only in the steps that consumed one
#define XPNG_DEC_RENORM(need, m)                                                          \
    do {                                                                                  \
        const uint32_t n1_ = (m) >> 1, n0_ = (m) & 1u;                                    \
        const uint32_t wsel_ = par ? w1 : (n1_ ? w2 : w1);                                \
        if (need) s = (s << 32) | wsel_;                                                  \
        rw = rw > n0_ + n1_ ? rw - (n0_ + n1_) : 0;                                       \
        XPNG_DEC_REFILL();                                                                \
        w1 = wring[(rw > 0 ? rw - 1 : 0) & 511u];                                         \
        w2 = wring[(rw > 1 ? rw - 2 : 0) & 511u];                                         \
    } while (0)
#define XPNG_DEC_PAIR()                                                                   \
    do {                                                                                  \
        i -= 2;                                                                           \
        uint32_t sym_ = 0;                                                                \
        bool need_;                                                                       \
        XPNG_DEC_ONE(sym_, need_);                                                        \
        const uint32_t m_ = sgpr((uint32_t)__ballot(need_) & 3u);                         \
        if (lane < 2) oring[(i + par) & 511u] = (uint8_t)sym_;                            \
        if (m_) XPNG_DEC_RENORM(need_, m_);                                               \
        if ((i & 511u) == 0) XPNG_DEC_FLUSH(i);                                           \
    } while (0)
    rw = sgpr(rw); ring_lo = sgpr(ring_lo);
    uint32_t i = sgpr(n);  // scalar
    if (n & 1) {  // odd tail comes from state0 only (libxpng.c:471-476)
        i--;
        uint32_t sym = 0;
        const uint64_t keep = s;
        bool need;
        XPNG_DEC_ONE(sym, need);
        if (par) s = keep;
        const uint32_t need0 = sgpr((uint32_t)__ballot(need && !par) & 1u);
        if (need0) {
            if (rw > 0) rw--;
            if (!par) s = (s << 32) | wring[rw & 511u];
        }
        if (lane == 0) oring[i & 511u] = (uint8_t)sym;
        if ((i & 511u) == 0) XPNG_DEC_FLUSH(i);
    }
    uint32_t w1 = wring[(rw > 0 ? rw - 1 : 0) & 511u], w2 = wring[(rw > 1 ? rw - 2 : 0) & 511u];
    while (i >= 2 && (i & 7u)) XPNG_DEC_PAIR();  // down to a multiple of 8 symbols
    // Main loop: straight-line blocks of 4 pair steps.  A lane packs its four symbols into one register; at the block end the
    // pair's 8 bytes are interleaved (DPP swap + two v_perm) and leave as ONE 8-byte LDS store, and the ring flush is checked
    // once: the per-step store / address arithmetic / loop bookkeeping (and the wait for that store at the top of every step)
    // are gone from the dependent chain.
    while (i >= 8) {
        uint32_t acc = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // (tried in round 3: the step first computed as if both states sat in a hot symbol and needed no refill, ONE ballot of
            //  "cold symbol OR refill" guarding the general path - 16.4 -> 16.7 ms on the 4096^2 image: on this alpha stream the pair
            //  is in the general path more than half of the steps, and those then pay the speculative work on top)
            uint32_t sym = 0;
            bool need;
            XPNG_DEC_ONE(sym, need);
            const uint32_t m = sgpr((uint32_t)__ballot(need) & 3u);
            if (m) XPNG_DEC_RENORM(need, m);
            acc = (acc << 8) | sym;
        }
        i -= 8;
        const uint32_t oth = swap_pair(acc);
        const uint32_t lo = __builtin_amdgcn_perm(oth, acc, 0x05010400u), hi = __builtin_amdgcn_perm(oth, acc, 0x07030602u);
        if (lane == 0) *reinterpret_cast<uint2 *>(oring + (i & 511u)) = make_uint2(lo, hi);
        if ((i & 511u) == 0) XPNG_DEC_FLUSH(i);
    }
    while (i >= 2) XPNG_DEC_PAIR();
#undef XPNG_DEC_PAIR
#undef XPNG_DEC_RENORM
#undef XPNG_DEC_ONE
#undef XPNG_DEC_FLUSH
#undef XPNG_DEC_REFILL
    XPNG_DSTAMP(2);
}

template <int MAXPB>
__global__ __launch_bounds__(64) void k_rans2_decode(const DecTile *__restrict__ info,
                                                     const TileDesc *__restrict__ tiles, TileSel sel, uint32_t c_first,
                                                     uint32_t c_count, int only_over, uint8_t *__restrict__ ctxsym,
                                                     uint8_t *__restrict__ asym, uint64_t *__restrict__ dbg) {
    rans2_decode_stream<MAXPB>(info, tiles, sel, blockIdx.x / c_count, c_first + blockIdx.x % c_count, only_over, ctxsym, asym, dbg);
}

// Wide mode: the streams k_rans2_dec_prep left to the narrow form (kind 2 among the context streams: never written by the
// reference).  A small fixed grid looks through the stream list 64 entries at a time - one workgroup per stream, as above, is
// 46 656 workgroups asking for 38 KB of LDS each to find nothing, on the critical path between the context chains and the walk.
template <int MAXPB>
__global__ __launch_bounds__(64) void k_rans2_decode_rest(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                          uint32_t total, uint32_t c_first, uint32_t c_count, uint8_t *__restrict__ ctxsym,
                                                          uint8_t *__restrict__ asym, uint64_t *__restrict__ dbg, const WDec *__restrict__ wdec) {
    const uint32_t lane = threadIdx.x & 63, n = total * c_count;
    for (uint32_t base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
        const uint32_t idx = base + lane;
        const bool mine = idx < n && wdec[(uint64_t)(idx / c_count) * 10 + c_first + idx % c_count].kind == 2;
        uint64_t m = __ballot(mine);
        while (m) {
            const uint32_t id2 = sgpr(base + (uint32_t)__builtin_ctzll(m));
            m &= m - 1;
            rans2_decode_stream<MAXPB, false>(info, tiles, sel, id2 / c_count, c_first + id2 % c_count, 2, ctxsym, asym, dbg);
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------------------
// alpha plane (RGBA).  a(x,y) = a(left) + d, except column 0: a(0,y) = a(0,y-1) + d (libxpng.c:798-800 with
// pr = p1x_ for rows 0 / interior and p1y_ for column 0).  grid = tiles, block = 1024.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_dec_alpha(const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, TileSel sel,
                                                    const uint8_t *__restrict__ asym, uint8_t *__restrict__ alpha) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DecTile d = info[j];
    if (d.type == 0 || d.type == TILE_BAD) return;
    const TileDesc t = tiles[vtile(sel, j)];
    const uint8_t *sy = asym + t.pbase;  // sy[i-1] = symbol of pixel i
    uint8_t *al = alpha + t.pbase;
    __shared__ uint32_t s_wave[THREADS / 64];
    __shared__ uint32_t s_carry;
    // first pixel's alpha: 4th byte of the first k word (MSB-first R,G,B,A)
    const uint32_t a0 = ld32u(d.blob + 8) & 0xFF;
    // ---- column 0: inclusive scan down the rows
    if (tid == 0) s_carry = a0;
    __syncthreads();
    if (tid == 0) al[0] = (uint8_t)a0;
    for (uint32_t y0 = 1; y0 < t.h; y0 += THREADS) {
        const uint32_t y = y0 + tid;
        const uint32_t dv = y < t.h ? (uint32_t)zz_dec(sy[(uint64_t)y * t.w - 1]) & 255u : 0u;
        uint32_t incl = dv;
        incl = wave_scan_incl(incl);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        uint32_t base = s_carry;
        for (uint32_t w2 = 0; w2 < wv; w2++) base += s_wave[w2];
        if (y < t.h) al[(uint64_t)y * t.w] = (uint8_t)(base + incl);
        __syncthreads();
        if (tid == THREADS - 1) s_carry = (base + incl) & 255u;
        __syncthreads();
    }
    __syncthreads();
    // ---- rows: each wave scans whole rows left to right, 256 pixels per step (a lane owns one aligned dword of the plane);
    // the symbol dwords and the row's column-0 value are read one step ahead, interior dwords are stored whole
    if (t.w == 1) return;
    constexpr uint32_t NWV = THREADS / 64;
    const uint32_t *sy4 = reinterpret_cast<const uint32_t *>(sy);
    uint32_t *al4 = reinterpret_cast<uint32_t *>(al);
    const uint32_t spr = ((t.w + 6) / 4 + 63) / 64;  // steps per row (rows start at any byte alignment)
    uint32_t y = wv, st = 0, carry = 0;
    uint32_t n_cur = 0, n_prev = 0, n_c0 = 0;
    if (y < t.h) {
        const uint32_t rs = y * t.w, g = (rs >> 2) + lane;
        if (4 * g < rs + t.w) { n_cur = sy4[g]; n_prev = g ? sy4[g - 1] : 0u; }
        n_c0 = al[rs];
    }
    while (y < t.h) {
        const uint32_t cur = n_cur, prev = n_prev, c0 = n_c0;
        const uint32_t rs = y * t.w, re = rs + t.w, g = (rs >> 2) + lane + 64 * st, idx0 = 4 * g;
        uint32_t ny = y, nst = st + 1;
        if (nst == spr) { nst = 0; ny = y + NWV; }
        if (ny < t.h) {
            const uint32_t nrs = ny * t.w, ng = (nrs >> 2) + lane + 64 * nst;
            if (4 * ng < nrs + t.w) { n_cur = sy4[ng]; n_prev = ng ? sy4[ng - 1] : 0u; }
            if (nst == 0) n_c0 = al[nrs];
        }
        if (st == 0) carry = c0;
        // deltas of pixels idx0 .. idx0+3 (symbol of pixel i sits at sy[i-1]); pixels outside (rs, re) contribute nothing
        const uint32_t u = __builtin_amdgcn_alignbyte(cur, prev, 3);
        uint32_t d4 = ((u >> 1) & 0x7F7F7F7Fu) ^ ((u & 0x01010101u) * 255u);
        const int32_t lo = (int32_t)rs - (int32_t)idx0 + 1, hi = (int32_t)re - (int32_t)idx0;  // bytes [lo, hi) of the dword are in the row
        uint32_t mask = lo <= 0 ? 0xFFFFFFFFu : (lo >= 4 ? 0u : 0xFFFFFFFFu << (8 * lo));
        mask &= hi >= 4 ? 0xFFFFFFFFu : (hi <= 0 ? 0u : ~(0xFFFFFFFFu << (8 * hi)));
        d4 &= mask;
        const uint32_t p0 = d4 & 255u, p1 = p0 + ((d4 >> 8) & 255u), p2 = p1 + ((d4 >> 16) & 255u), p3 = p2 + (d4 >> 24);
        const uint32_t incl = wave_scan_incl(p3);
        const uint32_t base = carry + incl - p3;
        const uint32_t b0 = (base + p0) & 255u, b1 = (base + p1) & 255u, b2 = (base + p2) & 255u, b3 = (base + p3) & 255u;
        if (mask == 0xFFFFFFFFu) al4[g] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        else if (mask) {
            if (mask & 0x000000FFu) al[idx0] = (uint8_t)b0;
            if (mask & 0x0000FF00u) al[idx0 + 1] = (uint8_t)b1;
            if (mask & 0x00FF0000u) al[idx0 + 2] = (uint8_t)b2;
            if (mask & 0xFF000000u) al[idx0 + 3] = (uint8_t)b3;
        }
        carry = (carry + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63)) & 255u;
        y = ny; st = nst;
    }
}

// --------------------------------------------------------------------------------------------------
// context chain (libxpng.c:803: nl = *cx[nl]++, starting from 0).  Strictly serial per tile, so one wave per tile and
// the step is made as short as the hardware allows: lane c < 9 owns queue c as an 8-symbol register window (low byte =
// head) backed by two prefetched 8-byte chunks; a step is v_readlane (head of the current queue -> SGPR), a predicated
// 8-bit shift on the owning lane, and a scalar pack of the output.  No LDS, no global load on the dependent chain.
// Output: nl sequence in coded-pixel order.      grid = tiles, block = 64.
// core of the walk: `base` = 256-byte aligned start of the tile's symbol area, qs = this lane's queue start inside it
// (lanes 0..8), total = number of symbols to produce.  Single-wave workgroup.
constexpr uint32_t WALK_RING = 512, WALK_UNIT = WALK_RING / 2;  // LDS ring of one context queue, and the unit it is refilled in
__device__ inline void ctx_walk(const uint8_t *__restrict__ base, uint32_t qs_in, uint32_t total, uint8_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    // queue supply: every queue is staged through its own 512-byte LDS ring in 256-byte units (coalesced copies by a quarter
    // of the wave, triggered at 8-step block boundaries when a queue's read position nears the end of what is staged).  (1 KB
    // rings until round 3: the walk waves of a pipelined batch held 9 KB of LDS each for their whole life.)
    __shared__ __align__(16) uint8_t qring[9][WALK_RING];
    const uint32_t qs = lane < 9 ? qs_in : 0;
    const uint32_t qsa = qs & ~15u;  // 16-byte aligned start of this lane's queue inside the tile's symbol area
    for (uint32_t c = 0; c < 9; c++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)qsa, (int)c);
        const uint4 *src = reinterpret_cast<const uint4 *>(base + a) + lane;
        if (lane < WALK_RING / 16) reinterpret_cast<uint4 *>(qring[c])[lane] = src[0];  // both units
    }
    __syncthreads();
    uint32_t filled = 2;                      // units staged for this lane's queue
    const uint32_t p0 = qs & 15u;             // ring position of the queue's first symbol
    const uint32_t ql = lane < 9 ? lane : 0;  // lanes >= 9 never pop; they alias queue 0 harmlessly
    uint64_t win = *reinterpret_cast<const uint64_t *>(&qring[ql][p0 & ~7u]) >> (8 * (p0 & 7u));
    uint32_t have = 8 - (p0 & 7u);
    uint32_t rdpos = (p0 & ~7u) + 8;          // next ring position to load into the register window
    uint32_t cur = 0;
    auto restage = [&]() {  // uniform entry; copies one more unit for every queue that is within 32 bytes of its staged end
        uint64_t m = __ballot(lane < 9 && rdpos + 32 >= filled * WALK_UNIT);
        if (m == 0) return;
        __syncthreads();
        while (m) {
            const int c = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)qsa, c);
            const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)filled, c);
            if (lane < WALK_UNIT / 16) {
                const uint4 v = (reinterpret_cast<const uint4 *>(base + a) + lane)[u * (WALK_UNIT / 16)];
                reinterpret_cast<uint4 *>(qring[c])[(u & 1u) * (WALK_UNIT / 16) + lane] = v;
            }
            if ((int)lane == c) filled++;
        }
        __syncthreads();
    };
    auto pop = [&]() -> uint32_t {  // returns the next nl (wave-uniform) and advances the owning queue
        const uint32_t sym = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)win, (int)cur) & 0xFFu;
        const bool mine = lane == cur;
        win >>= mine ? 8 : 0;
        have -= mine ? 1u : 0u;
        if (have == 0) {  // divergent, once per 8 pops of a queue: next 8 symbols from the LDS ring
            win = *reinterpret_cast<const uint64_t *>(&qring[ql][rdpos & (WALK_RING - 1)]);
            rdpos += 8;
            have = 8;
        }
        cur = sym;
        return sym;
    };
    uint32_t k = 0;
    for (; k + 8 <= total; k += 8) {
        restage();
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) lo |= pop() << (8 * u);
#pragma unroll
        for (int u = 0; u < 4; u++) hi |= pop() << (8 * u);
        if (lane == 0) *reinterpret_cast<uint2 *>(out + k) = make_uint2(lo, hi);
    }
    restage();
    for (; k < total; k++) {
        const uint32_t sy = pop();
        if (lane == 0) out[k] = (uint8_t)sy;
    }
}

// --------------------------------------------------------------------------------------------------
// The same chain on the SCALAR unit (single image: one wave per tile, latency is everything).  A step of the v_readlane form
// above is ~16 VALU instructions of which five are dependent (readlane -> s_and -> v_cmp -> v_cndmask -> 64-bit shift ->
// readlane), 45-60 ns.  Here the nine queue heads are 64-bit SGPR pairs s[40+2c : 41+2c] = (four symbols | marker bit 32) and the
// step is nine scalar instructions, ~21 ns:
//     m0 = 2 cur;  t = SGPR[40 + m0] (s_movrels_b64);  cur = t & 0xff;  t >>= 8;  t == 1 ? refill;  SGPR[40 + m0] = t;  out lane k = cur
// The refill (once per four pops of a queue) takes the queue's next dword from lane c's register pair (w0, w1 <- its LDS ring)
// by v_readlane; the LDS read that tops w1 up has eight pops of that queue to land.  The head registers live in VGPR lanes
// between blocks of 64 steps (the compiler may use any SGPR between two asm statements), symbols > 8 (corrupt streams only)
// are masked to 4 bits and index spare pairs s[58:71], so no register outside s[40:77] is ever touched.
// Requires 4-byte aligned queue starts (k_dec_parse aligns them to 16).
#define XPNG_WS(k)                                                                                                                  \
    "s_lshl_b32 m0, %[cur], 1\n s_nop 0\n s_movrels_b64 s[74:75], s[40:41]\n s_and_b32 %[cur], s74, 0xff\n s_lshr_b64 s[74:75], s[74:75], 8\n" \
    "s_cmp_eq_u64 s[74:75], 1\n s_cbranch_scc1 .Lwr%=_" #k "\n.Lwa%=_" #k ":\n s_movreld_b64 s[40:41], s[74:75]\n v_writelane_b32 %[vout], %[cur], " #k "\n"
#define XPNG_WR(k)                                                                                                                  \
    ".Lwr%=_" #k ":\n s_lshr_b32 s72, m0, 1\n s_nop 3\n v_readlane_b32 s74, %[w0], s72\n s_and_b32 s74, s74, 0x0f0f0f0f\n s_mov_b32 s75, 1\n"   \
    "v_cmp_eq_u32 vcc, s72, %[lanev]\n s_and_saveexec_b64 s[76:77], vcc\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %[w0], %[w1]\n"               \
    "v_and_b32 %[vt], 0x1ff, %[rd]\n v_add_u32 %[vt], %[vt], %[rbase]\n ds_read_b32 %[w1], %[vt]\n v_add_u32 %[rd], 4, %[rd]\n"           \
    "s_mov_b64 exec, s[76:77]\n s_branch .Lwa%=_" #k "\n"
__device__ inline void ctx_walk_salu(const uint8_t *__restrict__ base, uint32_t qs_in, uint32_t total, uint8_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    __shared__ __align__(16) uint8_t qring[9][WALK_RING];
    const uint32_t qs = lane < 9 ? qs_in : 0;
    const uint32_t qsa = qs & ~15u;
    for (uint32_t c = 0; c < 9; c++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)qsa, (int)c);
        if (lane < WALK_RING / 16) reinterpret_cast<uint4 *>(qring[c])[lane] = (reinterpret_cast<const uint4 *>(base + a) + lane)[0];  // both units
    }
    __syncthreads();
    uint32_t filled = 2;                      // units staged for this lane's queue
    const uint32_t ql = lane < 9 ? lane : 0;  // lanes >= 9 alias queue 0's ring; nothing they hold is ever a real symbol
    const uint32_t p0 = qs & 12u;
    const uint32_t *r0 = reinterpret_cast<const uint32_t *>(&qring[ql][p0]);
    uint32_t qlo = lane < 9 ? (r0[0] & 0x0F0F0F0Fu) : 0u, qhi = 1u;
    uint32_t w0 = r0[1], w1 = r0[2], rd = p0 + 12, vt, vout;
    const uint32_t rbase = (uint32_t)(uintptr_t)&qring[ql][0];  // LDS byte address of this lane's ring (low 32 bits of the flat address)
    uint32_t cur = 0;
    for (uint32_t k = 0; k < total; k += 64) {
        {   // stage one more unit for every queue whose reads can reach the end of what is staged within this block
            uint64_t m = __ballot(lane < 9 && rd + 96 >= filled * WALK_UNIT);
            if (m) {
                __syncthreads();
                while (m) {
                    const int c = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)qsa, c);
                    const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)filled, c);
                    if (lane < WALK_UNIT / 16) reinterpret_cast<uint4 *>(qring[c])[(u & 1u) * (WALK_UNIT / 16) + lane] = (reinterpret_cast<const uint4 *>(base + a) + lane)[u * (WALK_UNIT / 16)];
                    if ((int)lane == c) filled++;
                }
                __syncthreads();
            }
        }
        asm volatile(
            "s_mov_b32 s73, m0\n"
        "v_readlane_b32 s40, %[qlo], 0\n v_readlane_b32 s41, %[qhi], 0\n"
        "v_readlane_b32 s42, %[qlo], 1\n v_readlane_b32 s43, %[qhi], 1\n"
        "v_readlane_b32 s44, %[qlo], 2\n v_readlane_b32 s45, %[qhi], 2\n"
        "v_readlane_b32 s46, %[qlo], 3\n v_readlane_b32 s47, %[qhi], 3\n"
        "v_readlane_b32 s48, %[qlo], 4\n v_readlane_b32 s49, %[qhi], 4\n"
        "v_readlane_b32 s50, %[qlo], 5\n v_readlane_b32 s51, %[qhi], 5\n"
        "v_readlane_b32 s52, %[qlo], 6\n v_readlane_b32 s53, %[qhi], 6\n"
        "v_readlane_b32 s54, %[qlo], 7\n v_readlane_b32 s55, %[qhi], 7\n"
        "v_readlane_b32 s56, %[qlo], 8\n v_readlane_b32 s57, %[qhi], 8\n"
        "v_readlane_b32 s58, %[qlo], 9\n v_readlane_b32 s59, %[qhi], 9\n"
        "v_readlane_b32 s60, %[qlo], 10\n v_readlane_b32 s61, %[qhi], 10\n"
        "v_readlane_b32 s62, %[qlo], 11\n v_readlane_b32 s63, %[qhi], 11\n"
        "v_readlane_b32 s64, %[qlo], 12\n v_readlane_b32 s65, %[qhi], 12\n"
        "v_readlane_b32 s66, %[qlo], 13\n v_readlane_b32 s67, %[qhi], 13\n"
        "v_readlane_b32 s68, %[qlo], 14\n v_readlane_b32 s69, %[qhi], 14\n"
        "v_readlane_b32 s70, %[qlo], 15\n v_readlane_b32 s71, %[qhi], 15\n"
        "s_nop 3\n"
        XPNG_WS(0)
        XPNG_WS(1)
        XPNG_WS(2)
        XPNG_WS(3)
        XPNG_WS(4)
        XPNG_WS(5)
        XPNG_WS(6)
        XPNG_WS(7)
        XPNG_WS(8)
        XPNG_WS(9)
        XPNG_WS(10)
        XPNG_WS(11)
        XPNG_WS(12)
        XPNG_WS(13)
        XPNG_WS(14)
        XPNG_WS(15)
        XPNG_WS(16)
        XPNG_WS(17)
        XPNG_WS(18)
        XPNG_WS(19)
        XPNG_WS(20)
        XPNG_WS(21)
        XPNG_WS(22)
        XPNG_WS(23)
        XPNG_WS(24)
        XPNG_WS(25)
        XPNG_WS(26)
        XPNG_WS(27)
        XPNG_WS(28)
        XPNG_WS(29)
        XPNG_WS(30)
        XPNG_WS(31)
        XPNG_WS(32)
        XPNG_WS(33)
        XPNG_WS(34)
        XPNG_WS(35)
        XPNG_WS(36)
        XPNG_WS(37)
        XPNG_WS(38)
        XPNG_WS(39)
        XPNG_WS(40)
        XPNG_WS(41)
        XPNG_WS(42)
        XPNG_WS(43)
        XPNG_WS(44)
        XPNG_WS(45)
        XPNG_WS(46)
        XPNG_WS(47)
        XPNG_WS(48)
        XPNG_WS(49)
        XPNG_WS(50)
        XPNG_WS(51)
        XPNG_WS(52)
        XPNG_WS(53)
        XPNG_WS(54)
        XPNG_WS(55)
        XPNG_WS(56)
        XPNG_WS(57)
        XPNG_WS(58)
        XPNG_WS(59)
        XPNG_WS(60)
        XPNG_WS(61)
        XPNG_WS(62)
        XPNG_WS(63)
            "s_branch .Lwend%=\n"
        XPNG_WR(0)
        XPNG_WR(1)
        XPNG_WR(2)
        XPNG_WR(3)
        XPNG_WR(4)
        XPNG_WR(5)
        XPNG_WR(6)
        XPNG_WR(7)
        XPNG_WR(8)
        XPNG_WR(9)
        XPNG_WR(10)
        XPNG_WR(11)
        XPNG_WR(12)
        XPNG_WR(13)
        XPNG_WR(14)
        XPNG_WR(15)
        XPNG_WR(16)
        XPNG_WR(17)
        XPNG_WR(18)
        XPNG_WR(19)
        XPNG_WR(20)
        XPNG_WR(21)
        XPNG_WR(22)
        XPNG_WR(23)
        XPNG_WR(24)
        XPNG_WR(25)
        XPNG_WR(26)
        XPNG_WR(27)
        XPNG_WR(28)
        XPNG_WR(29)
        XPNG_WR(30)
        XPNG_WR(31)
        XPNG_WR(32)
        XPNG_WR(33)
        XPNG_WR(34)
        XPNG_WR(35)
        XPNG_WR(36)
        XPNG_WR(37)
        XPNG_WR(38)
        XPNG_WR(39)
        XPNG_WR(40)
        XPNG_WR(41)
        XPNG_WR(42)
        XPNG_WR(43)
        XPNG_WR(44)
        XPNG_WR(45)
        XPNG_WR(46)
        XPNG_WR(47)
        XPNG_WR(48)
        XPNG_WR(49)
        XPNG_WR(50)
        XPNG_WR(51)
        XPNG_WR(52)
        XPNG_WR(53)
        XPNG_WR(54)
        XPNG_WR(55)
        XPNG_WR(56)
        XPNG_WR(57)
        XPNG_WR(58)
        XPNG_WR(59)
        XPNG_WR(60)
        XPNG_WR(61)
        XPNG_WR(62)
        XPNG_WR(63)
            ".Lwend%=:\n"
        "v_writelane_b32 %[qlo], s40, 0\n v_writelane_b32 %[qhi], s41, 0\n"
        "v_writelane_b32 %[qlo], s42, 1\n v_writelane_b32 %[qhi], s43, 1\n"
        "v_writelane_b32 %[qlo], s44, 2\n v_writelane_b32 %[qhi], s45, 2\n"
        "v_writelane_b32 %[qlo], s46, 3\n v_writelane_b32 %[qhi], s47, 3\n"
        "v_writelane_b32 %[qlo], s48, 4\n v_writelane_b32 %[qhi], s49, 4\n"
        "v_writelane_b32 %[qlo], s50, 5\n v_writelane_b32 %[qhi], s51, 5\n"
        "v_writelane_b32 %[qlo], s52, 6\n v_writelane_b32 %[qhi], s53, 6\n"
        "v_writelane_b32 %[qlo], s54, 7\n v_writelane_b32 %[qhi], s55, 7\n"
        "v_writelane_b32 %[qlo], s56, 8\n v_writelane_b32 %[qhi], s57, 8\n"
            "s_mov_b32 m0, s73\n"
            : [qlo] "+v"(qlo), [qhi] "+v"(qhi), [w0] "+v"(w0), [w1] "+v"(w1), [rd] "+v"(rd), [vt] "=&v"(vt), [vout] "=&v"(vout), [cur] "+s"(cur)
            : [rbase] "v"(rbase), [lanev] "v"(lane)
            : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "vcc", "scc", "memory");
        if (k + lane < total) out[k + lane] = (uint8_t)vout;
    }
}
#undef XPNG_WS
#undef XPNG_WR

__global__ __launch_bounds__(64) void k_dec_walk(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles,
                                                 TileSel sel, const uint8_t *__restrict__ ctxsym,
                                                 uint8_t *__restrict__ nlseq) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x & 63;
    const DecTile d = info[j];
    if (d.type == 0 || d.type == TILE_BAD) return;
    const TileDesc t = tiles[vtile(sel, j)];
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.ctx_start[9]);
    XPNG_PROBE_BEGIN()
    ctx_walk_salu(ctxsym + t.pbase, lane < 9 ? d.ctx_start[lane] : 0, total, nlseq + t.pbase);  // (k_dec_parse aligns the queue starts to 16 bytes)
    XPNG_PROBE_END(6)
}

// --------------------------------------------------------------------------------------------------
// The same walk for large batches: one LANE per tile, 64 tiles per wave (k_dec_walk keeps a whole wave, and 9 KB of
// LDS, busy per tile; with thousands of tiles in flight that footprint, not the chain latency, is what hurts).
//   head[c][lane]  = { bytes left in the current dword of queue c (next symbol lowest), position, the following dword }
//   ringw[c][w][lane] = 64-byte window of queue c as 16 dwords, lane-minor so that 64 lanes never share a bank
// The dependent chain of a step is ONE LDS round trip: ds_read_b128 head[cur] -> symbol -> cur.  The bookkeeping of the
// previous step (advance the position, pull the next window dword, write the head back) is issued behind that read and
// completes while it is in flight; when the same queue is popped twice in a row the read is stale and the registers of
// the previous step are forwarded instead.  Queue 9 is a parking queue that returns 9 forever: a lane whose tile is
// finished walks it.  Global memory is touched only at 16-step block boundaries: aligned 16-byte chunks are requested for a
// queue at one service and land in its window at the next (the chunk starts are 16-byte aligned, k_dec_parse).
// LPW = tiles per wavefront (lanes 0 .. LPW - 1 carry one each, the others idle through the loop): the LDS of a wavefront is 800 LPW
// bytes - 51 KB at 64.  [r4] A workgroup that asks for 51 KB is placed only when a compute unit has that much free at once, and the
// chip-filling kernels of the other pipeline slots re-occupy every smaller hole at once: profiles/r03_wave_probe_p4.txt has the
// wide walk's wavefronts starting up to 6 (p90) / 10 ms (max) after the first one - on the decode's critical path, a kernel of ~20 ms
// of work spans ~30.  With 32 tiles per wavefront a workgroup asks for 25.6 KB (and waits for 288 scattered lines per service
// instead of 576); the price is twice the wavefronts, i.e. twice the walk's vector instructions.
constexpr uint32_t WALK_WIDE_COLS(uint32_t lpw) { return lpw + (lpw < 64 ? 1u : 0u); }  // LDS columns: one per tile + one that the idle lanes share
constexpr size_t WALK_WIDE_LDS_BYTES(uint32_t lpw) { return 10 * (size_t)WALK_WIDE_COLS(lpw) * 16 + 10 * 16 * (size_t)WALK_WIDE_COLS(lpw) * 4; }  // heads + windows (k_dec_walk_wide)
template <uint32_t LPW>
__global__ __launch_bounds__(64) void k_dec_walk_wide(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles,
                                                      TileSel sel, uint32_t total_tiles, const uint8_t *__restrict__ ctxsym,
                                                      uint8_t *__restrict__ nlseq, uint32_t j0) {
    // window: WCH chunks of 16 B per queue; a queue is serviced every SVC-th block.  (8 chunks / every 4th block is ~8 % faster
    // alone, but 90 KB of LDS per wave instead of 50 costs the kernels beside it more than that.)
    constexpr uint32_t WCH = 4, SVC = 2, WDW = WCH * 4, lpw = LPW, COLS = WALK_WIDE_COLS(LPW);
    static_assert(WALK_WIDE_LDS_BYTES(LPW) == 10 * COLS * 16 + 10 * WDW * COLS * 4 && LPW >= 8 && LPW <= 64, "LDS layout");
    extern __shared__ __align__(16) uint8_t walk_wide_lds[];  // dynamic (common.hpp: why)
    u32x4_t *const head = reinterpret_cast<u32x4_t *>(walk_wide_lds);                         // [10 * COLS]
    uint32_t *const ringw = reinterpret_cast<uint32_t *>(walk_wide_lds + 10 * COLS * 16);    // [10 * WDW * COLS]
    __builtin_amdgcn_s_setprio(XPNG_CHAIN_PRIO);  // a serial chain: its latency is the critical path, the throughput kernels beside it are not
    XPNG_PROBE_BEGIN()
    const uint32_t lane_ = threadIdx.x & 63, j = j0 + blockIdx.x * lpw + lane_;  // work items [j0, total_tiles)
    bool live = lane_ < lpw && j < total_tiles;
    const uint32_t lane = lane_ < lpw ? lane_ : lpw;  // LDS column of this lane (the idle lanes of a narrower form share the last one: they only ever walk the parking queue, whose heads are self-consistent whoever wrote them)
    const DecTile *d = info + (live ? j : 0);
    live = live && d->type != 0 && d->type != TILE_BAD;
    const TileDesc *t = tiles + vtile(sel, live ? j : 0);
    const uint32_t total = live ? d->ctx_start[9] : 0;
    const uintptr_t base = (uintptr_t)(ctxsym + t->pbase);
    uint8_t *out = nlseq + t->pbase;
    // where a lane without (more) symbols stores: behind the nl sequence of the tile its `out` points at (the planes carry >= 192
    // bytes of slack behind a tile; a lane without a tile points at its placeholder tile, whose sequence another lane may be writing)
    const DecTile *dh = info + (live ? j : 0);  // the tile `out` points at
    const uint32_t seq_here = (dh->type != 0 && dh->type != TILE_BAD) ? dh->ctx_start[9] : 0u;
    const uint32_t dump = (seq_here + 15u) & ~15u;
    uint32_t qoff[9], have[9];
    u32x4_t fl[9][SVC];  // in flight per queue: chunks have .. have + SVC - 1
    typedef const __attribute__((address_space(1))) u32x4_t *gp128;
    const u32x4_t zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 9; c++) {
        qoff[c] = live ? d->ctx_start[c] : 0;
        const gp128 src = (gp128)(base + qoff[c]);
#pragma unroll
        for (int q = 0; q < (int)SVC; q++) {
            u32x4_t v = zero4;
            if (live) v = src[q];
            ringw[(c * WDW + q * 4 + 0) * COLS + lane] = v.x; ringw[(c * WDW + q * 4 + 1) * COLS + lane] = v.y;
            ringw[(c * WDW + q * 4 + 2) * COLS + lane] = v.z; ringw[(c * WDW + q * 4 + 3) * COLS + lane] = v.w;
            if (q == 0) { const u32x4_t h = {v.x, 0u, v.y, 0u}; head[c * COLS + lane] = h; }
        }
        have[c] = SVC;
#pragma unroll
        for (int q = 0; q < (int)SVC; q++) { fl[c][q] = zero4; if (live) fl[c][q] = src[SVC + q]; }
    }
#pragma unroll
    for (int w = 0; w < (int)WDW; w++) ringw[(9 * WDW + w) * COLS + lane] = 0x09090909u;
    { const u32x4_t h = {0x09090909u, 0u, 0x09090909u, 0u}; head[9 * COLS + lane] = h; }
    uint32_t T = total;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(T, o); T = v > T ? v : T; }
    T = sgpr(T);
    uint32_t cur = total > 0 ? 0u : 9u;
    // registers of the previous step: its queue and the head it popped from (lo, pos, nxt); queue 9 at start (harmless)
    uint32_t pcur = 9, plo = 0x09090909u, ppos = 0, pnxt = 0x09090909u;
    auto block = [&](uint32_t kb) __attribute__((always_inline)) {
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; u++) {
            // bookkeeping of the previous step: its window read goes out first, then this step's head read; both are in
            // flight together
            const uint32_t npos = ppos + 1;
            const bool cross = (npos & 3u) == 0;
            const uint32_t rr = ringw[(pcur * WDW + (((npos >> 2) + 1) & (WDW - 1))) * COLS + lane];
            const u32x4_t r = head[cur * COLS + lane];               // (stale if cur == pcur: forwarded below)
            const uint32_t nlo = cross ? pnxt : plo >> 8, nnxt = cross ? rr : pnxt;
            { const u32x4_t h = {nlo, npos, nnxt, 0u}; head[pcur * COLS + lane] = h; }
            const bool same = cur == pcur;
            plo = same ? nlo : r.x; ppos = same ? npos : r.y; pnxt = same ? nnxt : (r.z | r.w);  // (r.w == 0; keeps the 4th register of the read alive so nothing else is loaded into it early)
            pcur = cur;
            const uint32_t sym = plo & 255u;
            o[u >> 2] |= sym << (8 * (u & 3));
            cur = kb + (uint32_t)u + 1 >= total ? 9u : (sym < 9u ? sym : 9u);  // (symbols > 8 only in corrupt streams)
        }
        // (an UNCONDITIONAL store: a lane past the end of its sequence writes its 16 bytes into the slack behind it.  Loads and stores
        //  retire in order on gfx9; with a store that may or may not have been issued the wait in front of a landing has to assume it was
        //  not, and then retires one load too many - one that is only a block old.)
        *reinterpret_cast<uint4 *>(out + (kb < total ? kb : dump)) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    // Service of the queues of one phase, every SVC-th block boundary (so a request has 16 * SVC steps, not 16, to come back:
    // 64 lanes x 9 queues are 576 different cache lines per round).  Up to SVC
    // chunks land while their window slots have been read out; after a service the window holds >= 16 * SVC unread bytes
    // (SVC blocks' worth) or everything up to its capacity; the next SVC missing chunks are requested again either way.
    // (The head of queue pcur in LDS is one pop behind the registers; a chunk index can only be underestimated by that.)
    auto service = [&](int phase) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 9; c++) {
            if ((c & (int)(SVC - 1)) != phase) continue;
            const uint32_t cq = head[c * COLS + lane].y >> 4;  // chunk the queue is reading from
#pragma unroll
            for (int q = 0; q < (int)SVC; q++) {
                if (have[c] - cq < WCH) {  // (once one chunk does not fit, neither do the later ones: have stops growing)
                    const uint32_t w0 = (c * WDW + (have[c] & (WCH - 1)) * 4) * COLS + lane;
                    ringw[w0] = fl[c][q].x; ringw[w0 + COLS] = fl[c][q].y; ringw[w0 + 2 * COLS] = fl[c][q].z; ringw[w0 + 3 * COLS] = fl[c][q].w;
                    have[c]++;
                }
            }
            {   // (unconditional: a branch around these loads makes the compiler copy the loaded registers into the loop-carried
                //  ones, and a copy is a use - it waits for the loads it has just issued; lanes without a tile read their
                //  placeholder tile's bytes, which nobody looks at)
                const gp128 src = (gp128)(base + qoff[c]) + have[c];
#pragma unroll
                for (int q = 0; q < (int)SVC; q++) fl[c][q] = src[q];
            }
        }
    };
    // Two blocks per loop iteration, each followed by the service of ITS phase as a compile-time constant: every in-flight
    // register set is then written at one fixed place of the loop body, so no register copies (= early waits) are needed
    // and the wait in front of a landing counts only what was issued before it.
    static_assert(SVC == 2, "the loop body below is written for two service phases");
#pragma unroll 1
    for (uint32_t kb = 0; kb < T; kb += 32) {
        block(kb);
        service(0);
        block(kb + 16);
        service(1);
    }
    XPNG_PROBE_END(5)
}

// --------------------------------------------------------------------------------------------------
// residual extraction.  Walks the tile's pixels in raster order, 1024 per step: coded flag (alpha != 0),
// coded index = running count, bit cursor = running sum of 3*nl; pulls 3*nl bits out of k, undoes zig-zag
// and the green subtraction, and stores one packed word per pixel: r | g<<8 | b<<16 | coded<<24.
//
// FOLD [r4] (RGBA, every tile at least 4 pixels wide): the alpha plane is never materialised.  Through round 3 k_dec_alpha turned the
// alpha symbols into a plane (1 B/px read, 1 B/px written, 2.6 B/px by the counters) that this kernel read back; now it takes the
// symbols themselves.  alpha(x, y) = alpha(0, y) + sum of the row's deltas up to x, alpha(0, y) = a0 + sum of column 0's deltas up to
// y (libxpng.c:798-800: left neighbour everywhere except column 0), all mod 256.  A prologue scans column 0 (h strided symbols) into
// the tile's first h bytes of the `alpha` buffer (scratch now); in the raster-order walk every row start is a "head" whose alpha is
// that absolute value, every other pixel adds its delta to a running sum P that is NEVER reset:
//     alpha(pixel) = P(pixel) + H,   H = (alpha - P) at the latest head at or before the pixel
// so one plain prefix sum (P) and one "latest head" propagation (a running maximum of (lane + 1) << 8 | H over the wave) do it; across
// waves and iterations two bytes are carried (P and H).  One more barrier per iteration than the plane-reading form.
template <int PXSZ, int THREADS, bool FOLD = false>
__global__ __launch_bounds__(THREADS) void k_dec_resid(const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, TileSel sel,
                                                    uint8_t *__restrict__ alpha, const uint8_t *__restrict__ asym, const uint8_t *__restrict__ nlseq,
                                                    uint32_t *__restrict__ resid, uint32_t j0) {
    bw_prio();
    static_assert(!FOLD || PXSZ == 4, "only RGBA tiles carry alpha");
    // One workgroup walks a tile in raster order, THREADS * 4 pixels per iteration; a lane owns 4 consecutive pixels: one
    // dword of the alpha plane (read one iteration ahead), up to 4 consecutive nl symbols, up to 96 bits of k, one 16-byte
    // store of residual words.  Two wave scans (coded pixels, bit lengths) and two barriers per iteration.
    const uint32_t j = j0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DecTile d = info[j];
    if (d.type == 0 || d.type == TILE_BAD) return;
    const TileDesc t = tiles[vtile(sel, j)];
    const uint8_t *kbase = d.blob + 8;
    const uint32_t kmis = (uint32_t)(reinterpret_cast<uintptr_t>(kbase) & 3);  // tile blobs are only byte-aligned after a raw RGB tile
    const uint32_t *kal = reinterpret_cast<const uint32_t *>(kbase - kmis);
    const uint32_t kwords = (d.kbytes + kmis + 3) >> 2;  // aligned dwords that hold k bytes
    const uint32_t *al = reinterpret_cast<const uint32_t *>(alpha + t.pbase);
    const uint32_t *nls = reinterpret_cast<const uint32_t *>(nlseq + t.pbase);
    uint32_t *rs = resid + t.pbase;
    const bool useG = d.type & 1;
    constexpr uint32_t NW = THREADS / 64, PX = THREADS * 4;
    __shared__ uint32_t s_wc[2][NW], s_wb[2][NW];
    __shared__ uint32_t s_wa[2][NW];  // FOLD: per wave: sum of its deltas | H of its last head << 8 | "has a head" << 16
    __shared__ uint32_t s_colw[NW], s_colc;
    uint32_t run_cnt = 0, run_bits = 8 * PXSZ, par = 0;
    // (x, y) of the lane's first pixel, advanced by PX pixels per iteration (the green add-back and the row starts of FOLD need it)
    uint32_t py = (4 * tid) / t.w, px = 4 * tid - py * t.w;
    const uint32_t dy = PX / t.w, dx = PX - dy * t.w;

    // ---- FOLD: column 0 into c0[0 .. h), then the symbol dwords and the head value of the first iteration
    const uint8_t *sy = FOLD ? asym + t.pbase : nullptr;               // sy[i - 1] = symbol of pixel i
    const uint32_t *sy4 = reinterpret_cast<const uint32_t *>(sy);
    uint8_t *c0 = alpha + t.pbase;
    uint32_t carryP = 0, carryH = 0, nx_cur = 0, nx_prev = 0, nx_A = 0;
    // head of a lane whose first pixel is (x, y) with tile index i: position hk inside the lane's four pixels and its row
    auto head_of = [&](uint32_t x, uint32_t y, uint32_t i, uint32_t &hk, uint32_t &hy) __attribute__((always_inline)) -> bool {
        hk = x == 0 ? 0u : t.w - x; hy = x == 0 ? y : y + 1;
        return (x == 0 || x + 4 > t.w) && i + hk < t.n;
    };
    if (FOLD) {
        const uint32_t a0 = ld32u(d.blob + 8) & 0xFF;  // first pixel's alpha: 4th byte of the first k word (MSB-first R,G,B,A)
        if (tid == 0) { s_colc = a0; c0[0] = (uint8_t)a0; }
        __syncthreads();
        for (uint32_t y0 = 1; y0 < t.h; y0 += THREADS) {
            const uint32_t y = y0 + tid;
            const uint32_t dv = y < t.h ? (uint32_t)zz_dec(sy[(uint64_t)y * t.w - 1]) & 255u : 0u;
            const uint32_t incl = wave_scan_incl(dv);
            if (lane == 63) s_colw[wv] = incl;
            __syncthreads();
            uint32_t base = s_colc;
            for (uint32_t w2 = 0; w2 < wv; w2++) base += s_colw[w2];
            if (y < t.h) c0[y] = (uint8_t)(base + incl);
            __syncthreads();
            if (tid == THREADS - 1) s_colc = (base + incl) & 255u;
            __syncthreads();
        }
        __threadfence_block();  // c0[] is read back below by other waves of this workgroup
        __syncthreads();
        if (4 * tid < t.n) {
            nx_cur = sy4[tid]; nx_prev = tid ? sy4[tid - 1] : 0u;
            uint32_t hk, hy;
            if (head_of(px, py, 4 * tid, hk, hy)) nx_A = c0[hy];
        }
    }

    uint32_t nx_a = 0;
    if (PXSZ == 4 && !FOLD && 4 * tid < t.n) nx_a = al[tid];
    for (uint32_t i0 = 0; i0 < t.n; i0 += PX, par ^= 1) {
        const uint32_t i = i0 + 4 * tid;
        uint32_t a4 = PXSZ == 4 ? nx_a : 0x01010101u;
        if (PXSZ == 4 && !FOLD) { const uint32_t in = i + PX; nx_a = 0; if (in < t.n) nx_a = al[in >> 2]; }
        if (FOLD) {
            const uint32_t cur = nx_cur, prev = nx_prev, A = nx_A;
            uint32_t hk, hy;
            const bool has_head = head_of(px, py, i, hk, hy);
            {   // the next iteration's symbol dwords and head value
                const uint32_t in = i + PX;
                uint32_t npx = px + dx, npy = py + dy;
                if (npx >= t.w) { npx -= t.w; npy++; }
                nx_cur = 0; nx_prev = 0; nx_A = 0;
                if (in < t.n) {
                    nx_cur = sy4[in >> 2]; nx_prev = sy4[(in >> 2) - 1];
                    uint32_t nhk, nhy;
                    if (head_of(npx, npy, in, nhk, nhy)) nx_A = c0[nhy];
                }
            }
            // deltas of pixels i .. i+3 (symbol of pixel i sits at sy[i-1]); a head and the pixels past the tile add nothing to P
            const uint32_t u = __builtin_amdgcn_alignbyte(cur, prev, 3);
            uint32_t d4 = ((u >> 1) & 0x7F7F7F7Fu) ^ ((u & 0x01010101u) * 255u);
            if (i >= t.n) d4 = 0;
            else if (t.n - i < 4) d4 &= 0xFFFFFFFFu >> (8 * (4 - (t.n - i)));
            if (has_head) d4 &= ~(0xFFu << (8 * hk));
            const uint32_t p0 = d4 & 255u, p1 = p0 + ((d4 >> 8) & 255u), p2 = p1 + ((d4 >> 16) & 255u), p3 = p2 + (d4 >> 24);
            const uint32_t incl = wave_scan_incl(p3);
            const uint32_t pb_ = incl - p3;  // wave-local P in front of this lane's first pixel
            const uint32_t pbefore = hk == 0 ? 0u : (hk == 1 ? p0 : (hk == 2 ? p1 : p2));  // the lane's deltas in front of its head
            const uint32_t Hown = (A - (pb_ + pbefore)) & 255u;
            const uint32_t M = wave_scan_max(has_head ? ((lane + 1u) << 8) | Hown : 0u);
            const uint32_t Mex = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);  // latest head in the lanes below
            if (lane == 63) s_wa[par][wv] = (incl & 255u) | ((M & 255u) << 8) | (M ? 1u << 16 : 0u);
            __syncthreads();
            uint32_t B = carryP, H = carryH, Cin = 0;
            for (uint32_t w2 = 0; w2 < NW; w2++) {
                const uint32_t e = s_wa[par][w2];
                if (w2 == wv) Cin = B + H;            // no head in this wave before the pixel: P and H both come from outside
                if (e >> 16) H = ((e >> 8) & 255u) - B;  // (a wave publishes H relative to its own P: B, the P in front of it, cancels inside the wave)
                B += e & 255u;
            }
            carryP = B & 255u; carryH = H & 255u;
            const uint32_t before = Mex ? Mex & 255u : Cin;  // H for the pixels in front of this lane's own head (or all four)
            const uint32_t pk[4] = {p0, p1, p2, p3};
            a4 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t off = (has_head && (uint32_t)k >= hk) ? Hown : before;
                a4 |= ((pb_ + pk[k] + off) & 255u) << (8 * k);
            }
        }
        if (i >= t.n) a4 = 0;
        else if (t.n - i < 4) a4 &= 0xFFFFFFFFu >> (8 * (4 - (t.n - i)));  // pixels past the tile
        if (i == 0) a4 &= 0xFFFFFF00u;                                      // the first pixel is not coded
        bool coded[4];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { coded[k] = ((a4 >> (8 * k)) & 255u) != 0; cnt += coded[k]; }
        const uint32_t cincl = wave_scan_incl(cnt);
        if (lane == 63) s_wc[par][wv] = cincl;
        __syncthreads();
        uint32_t cbase = run_cnt, ctot = 0;
        for (uint32_t w2 = 0; w2 < NW; w2++) { const uint32_t v = s_wc[par][w2]; if (w2 < wv) cbase += v; ctot += v; }
        // the lane's nl symbols: cnt consecutive bytes of the nl sequence
        uint32_t u = 0;
        if (cnt) {
            const uint32_t s0 = cbase + cincl - cnt;
            const uint32_t lo = nls[s0 >> 2], hi = nls[(s0 >> 2) + 1];
            u = __builtin_amdgcn_alignbyte(hi, lo, s0 & 3);
        }
        uint32_t nl[4], len[4], lane_len = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nl[k] = coded[k] ? min(u & 255u, 8u) : 0u;
            u = coded[k] ? u >> 8 : u;
            len[k] = 3 * nl[k];
            lane_len += len[k];
        }
        const uint32_t bincl = wave_scan_incl(lane_len);
        if (lane == 63) s_wb[par][wv] = bincl;
        __syncthreads();
        uint32_t bbase = run_bits, btot = 0;
        for (uint32_t w2 = 0; w2 < NW; w2++) { const uint32_t v = s_wb[par][w2]; if (w2 < wv) bbase += v; btot += v; }
        if (i < t.n) {
            // bits [eb - lane_len, eb) of k, right-aligned in s2:s1:s0 (<= 96 bits): the four k words that end at bit eb
            uint32_t s0 = 0, s1 = 0, s2 = 0;
            if (lane_len) {
                const uint32_t eb = bbase + bincl;
                const int32_t we = (int32_t)((eb - 1) >> 5);
                uint32_t A[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {  // aligned dwords we-3 .. we+1 of the (possibly misaligned) word array
                    const int32_t q = we - 3 + k;
                    A[k] = (q >= 0 && (uint32_t)q < kwords) ? kal[q] : 0u;
                }
                uint32_t Bw[4];
#pragma unroll
                for (int k = 0; k < 4; k++) Bw[k] = __builtin_amdgcn_alignbyte(A[k + 1], A[k], kmis);
                const uint32_t r = (0u - eb) & 31u;  // bits of the last word behind the lane's string
                s0 = __builtin_amdgcn_alignbit(Bw[2], Bw[3], r);
                s1 = __builtin_amdgcn_alignbit(Bw[1], Bw[2], r);
                s2 = __builtin_amdgcn_alignbit(Bw[0], Bw[1], r);
            }
            uint32_t word[4];
#pragma unroll
            for (int k = 3; k >= 0; k--) {  // last pixel first: its bits are the lowest
                const uint32_t v = __builtin_amdgcn_ubfe(s0, 0, len[k]);
                s0 = __builtin_amdgcn_alignbit(s1, s0, len[k]);
                s1 = __builtin_amdgcn_alignbit(s2, s1, len[k]);
                s2 >>= len[k];
                const uint32_t mk = (1u << nl[k]) - 1;
                int dr = zz_dec((int)(v >> (2 * nl[k]))), dg = zz_dec((int)((v >> nl[k]) & mk)), db = zz_dec((int)(v & mk));
                if (useG) {  // green add-back everywhere but row 0 / column 0 (libxpng.c:813)
                    uint32_t x = px + k, y = py;
                    if (x >= t.w) { x -= t.w; y++; }
                    if (x > 0 && y > 0) { dr += dg; db += dg; }
                }
                // top byte: "coded" marker = the pixel's alpha for RGBA (non-zero exactly when coded), 1 for RGB
                const uint32_t w = ((uint32_t)dr & 255u) | (((uint32_t)dg & 255u) << 8) | (((uint32_t)db & 255u) << 16) | (((a4 >> (8 * k)) & 255u) << 24);
                word[k] = coded[k] ? w : 0u;
            }
            *reinterpret_cast<u32x4_t *>(rs + i) = u32x4_t{word[0], word[1], word[2], word[3]};
        }
        run_cnt += ctot; run_bits += btot;
        px += dx; py += dy;
        if (px >= t.w) { px -= t.w; py++; }
    }
}

// --------------------------------------------------------------------------------------------------
// reconstruction.  Raw tiles: row copy (libxpng.c:846).  Coded tiles: pixel (x,y) needs reconstructed L, U, UL, so
// rows advance as an anti-diagonal wavefront: thread r handles row yb+r and, at step s, column s-r.  Its U is what
// thread r-1 produced one step earlier (LDS, double-buffered by step parity), its UL is its previous U, its L its own
// previous output.  Tiles taller than 1024 rows run in bands; a band's first row reads U from the raster.
// Anti-diagonal wavefront over one tile.  predmode: 0 = average (p2a), 1 = gradient (p3a), 2 = left (p1x), 3 = up (p1y) for
// interior pixels; row 0 always predicts from the left and column 0 from above.  `first` = the tile's first pixel.
// rs[i] = packed residual word of pixel i (r | g<<8 | b<<16 | coded<<24); al = alpha plane (RGBA) or unused.
template <int PXSZ>
__device__ inline void recon_wavefront(const TileDesc &t, uint8_t *__restrict__ dst, uint64_t bpr, const uint8_t *__restrict__ al,
                                       const uint32_t *__restrict__ rs, uint32_t first, int predmode, uint32_t (*s_row)[1024]) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t yb = 0; yb < t.h; yb += 1024) {
        const uint32_t rows = (t.h - yb) < 1024u ? (t.h - yb) : 1024u;
        const uint32_t y = yb + tid;
        const bool active = tid < rows;
        uint32_t L = 0, U = 0, UL = 0;
        const uint32_t steps = t.w + rows - 1;
        uint32_t nres = 0;  // residual word prefetched for the next step of this thread
        if (active && tid == 0) nres = rs[(uint64_t)y * t.w];
        for (uint32_t s = 0; s < steps; s++) {
            const int32_t x = (int32_t)s - (int32_t)tid;
            const bool on = active && x >= 0 && x < (int32_t)t.w;
            // U for this step: previous row's output at column x
            uint32_t Unew = 0;
            if (on && y > 0) {
                if (tid > 0) Unew = s_row[(s + 1) & 1][tid - 1];  // written at step s-1
                else Unew = load_px<PXSZ>(dst + (uint64_t)(y - 1) * bpr + (uint64_t)x * PXSZ);  // band seam: from the raster
            }
            uint32_t outpx = 0;
            if (on) {
                UL = U; U = Unew;
                const uint32_t rw = nres;
                const uint32_t i = y * t.w + (uint32_t)x;
                if (i == 0) outpx = first;
                else {
                    const bool coded = (rw >> 24) != 0;
                    if (coded) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const int l = (L >> (8 * c)) & 255, u = (U >> (8 * c)) & 255, ul = (UL >> (8 * c)) & 255;
                            int pred;
                            if (y == 0) pred = l;
                            else if (x == 0) pred = u;
                            else pred = predmode == 0 ? pred_avg(l, u) : predmode == 1 ? pred_grad(l, u, ul) : predmode == 2 ? l : u;
                            outpx |= (((rw >> (8 * c)) + (uint32_t)pred) & 255u) << (8 * c);
                        }
                    }
                    if (PXSZ == 4) outpx |= rw & 0xFF000000u;  // alpha travels in the residual word; alpha==0 -> whole pixel 0 (libxpng.c:802)
                }
                L = outpx;
                uint8_t *o = dst + (uint64_t)y * bpr + (uint64_t)x * PXSZ;
                if (PXSZ == 4) *reinterpret_cast<uint32_t *>(o) = outpx;
                else { o[0] = (uint8_t)outpx; o[1] = (uint8_t)(outpx >> 8); o[2] = (uint8_t)(outpx >> 16); }
                s_row[s & 1][tid] = outpx;
            }
            // prefetch the residual of the next step (column x+1 of this row)
            const int32_t xn = x + 1;
            if (active && xn >= 0 && xn < (int32_t)t.w) nres = rs[(uint64_t)y * t.w + (uint32_t)xn];
            __syncthreads();
        }
        __syncthreads();  // the band's last row is in the raster (global) before the next band reads it
        __threadfence_block();
    }
}

// Barrier-free form of the wavefront (tiles up to 1024 rows).  Wave w owns rows 64w..64w+63, lane l handles column s - l at
// its own step s; no workgroup barrier: a wave only waits for the wave above through a progress counter in LDS.
//   * U (row above, same column) is what lane l-1 produced one step earlier: one DPP wave_shr, no LDS;
//   * lane 0 takes U from the boundary row of the wave above, which that wave's last lane streams into an LDS row buffer
//     (64 columns are fetched at a time and handed out with v_readlane);
//   * residual words are prefetched 4 steps ahead into rotating registers.
// dynamic LDS: nw * ew dwords of boundary rows + 16 progress counters.
template <int PXSZ>
__device__ inline void recon_free(const TileDesc &t, uint8_t *__restrict__ dst, uint64_t bpr, const uint8_t *__restrict__ al,
                                  const uint32_t *__restrict__ rs, uint32_t first, int predmode, uint32_t *edge, uint32_t ew,
                                  uint32_t *prog) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t nw = (t.h + 63) >> 6;
    if (tid < 16) prog[tid] = 0;
    __syncthreads();
    if (w >= nw) return;
    const uint32_t y = 64 * w + lane;
    const bool active = y < t.h;
    const uint32_t *rsrow = rs + (uint64_t)(active ? y : 0) * t.w;
    (void)al;
    uint8_t *drow = dst + (uint64_t)(active ? y : 0) * bpr;
    const uint32_t S = t.w + 63;  // steps of this wave
    uint32_t *my_edge = edge + w * ew;
    const uint32_t *up_edge = edge + (w ? w - 1 : 0) * ew;
    const bool writer = lane == 63 && w + 1 < nw;  // (the last lane of a non-final wave is always an existing row)
    uint32_t prev = 0, U = 0, UL = 0, ev = 0;
    auto fetch = [&](uint32_t s) -> uint32_t {  // residual word this lane needs at step s
        const int32_t x = (int32_t)s - (int32_t)lane;
        return (active && x >= 0 && x < (int32_t)t.w) ? rsrow[x] : 0u;
    };
    uint32_t r0 = fetch(0), r1 = fetch(1), r2 = fetch(2), r3 = fetch(3);
    auto step = [&](uint32_t s, uint32_t rw) {
        if (w > 0 && (s & 63u) == 0 && s < t.w) {  // uniform: next 64 columns of the boundary row above
            const uint32_t need = s + 64 < t.w ? s + 64 : t.w;
            while (__hip_atomic_load(&prog[w - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(2);
            ev = s + lane < t.w ? up_edge[s + lane] : 0u;
        }
        // previous-step output of the lane above (lane 0: boundary row of the wave above)
        uint32_t Unew = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)prev, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)(s & 63u));
        if (lane == 0) Unew = e0;
        const int32_t x = (int32_t)s - (int32_t)lane;
        const bool on = active && x >= 0 && x < (int32_t)t.w;
        uint32_t outpx = 0;
        if (on) {
            UL = U; U = Unew;
            const uint32_t i = y * t.w + (uint32_t)x;
            if (i == 0) outpx = first;
            else {
                if ((rw >> 24) != 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int l = (prev >> (8 * c)) & 255, u = (U >> (8 * c)) & 255, ul = (UL >> (8 * c)) & 255;
                        int pred;
                        if (y == 0) pred = l;
                        else if (x == 0) pred = u;
                        else pred = predmode == 0 ? pred_avg(l, u) : predmode == 1 ? pred_grad(l, u, ul) : predmode == 2 ? l : u;
                        outpx |= (((rw >> (8 * c)) + (uint32_t)pred) & 255u) << (8 * c);
                    }
                }
                if (PXSZ == 4) outpx |= rw & 0xFF000000u;  // alpha travels in the residual word
            }
            uint8_t *o = drow + (uint64_t)x * PXSZ;
            if (PXSZ == 4) *reinterpret_cast<uint32_t *>(o) = outpx;
            else { o[0] = (uint8_t)outpx; o[1] = (uint8_t)(outpx >> 8); o[2] = (uint8_t)(outpx >> 16); }
            if (writer) {
                my_edge[x] = outpx;
                if ((x & 15) == 15 || x + 1 == (int32_t)t.w) __hip_atomic_store(&prog[w], (uint32_t)x + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            prev = outpx;
        }
    };
    uint32_t s = 0;
    for (; s + 4 <= S; s += 4) {
        uint32_t c0 = r0; r0 = fetch(s + 4); step(s, c0);
        uint32_t c1 = r1; r1 = fetch(s + 5); step(s + 1, c1);
        uint32_t c2 = r2; r2 = fetch(s + 6); step(s + 2, c2);
        uint32_t c3 = r3; r3 = fetch(s + 7); step(s + 3, c3);
    }
    if (s < S) { step(s, r0); s++; }
    if (s < S) { step(s, r1); s++; }
    if (s < S) { step(s, r2); s++; }
}

template <int PXSZ>
__global__ __launch_bounds__(1024) void k_dec_recon(const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, TileSel sel,
                                                    const uint8_t *__restrict__ alpha, const uint32_t *__restrict__ resid,
                                                    uint8_t *const *__restrict__ rasters, uint64_t bpr, uint32_t free_ew) {
    bw_prio();
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const DecTile d = info[j];
    const TileDesc t = tiles[vtile(sel, j)];
    uint8_t *dst = rasters[t.img] + (uint64_t)t.y * bpr + (uint64_t)t.x * PXSZ;
    if (d.type == TILE_BAD) return;
    if (d.type == 0) {
        const uint8_t *src = d.blob + 4;
        const uint64_t row = (uint64_t)t.w * PXSZ;
        for (uint64_t b = tid; b < row * t.h; b += blockDim.x) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = src[b]; }
        return;
    }
    // first pixel from the head of k (libxpng.c:850): bytes MSB-first
    const uint32_t kw0 = ld32u(d.blob + 8);
    uint32_t first = ((kw0 >> 24) & 255u) | (((kw0 >> 16) & 255u) << 8) | (((kw0 >> 8) & 255u) << 16);
    if (PXSZ == 4) first |= (kw0 & 255u) << 24;
    if (free_ew) {
        extern __shared__ uint32_t dyn_lds[];
        recon_free<PXSZ>(t, dst, bpr, alpha + t.pbase, resid + t.pbase, first, (d.type >> 1) & 1, dyn_lds + 16, free_ew, dyn_lds);
    } else {
        __shared__ uint32_t s_row[2][1024];
        recon_wavefront<PXSZ>(t, dst, bpr, alpha + t.pbase, resid + t.pbase, first, (d.type >> 1) & 1, s_row);
    }
}

// --------------------------------------------------------------------------------------------------
// Reconstruction for large batches: ONE wavefront per tile (k_dec_recon spreads a tile over up to 9 waves that hand rows
// to each other through LDS and progress counters: right for one image, but with thousands of tiles in flight the
// per-step hand-over latency, not the arithmetic, sets the pace).  The tile is swept in bands of 64 rows; lane r owns row
// yb + r and at step s reconstructs column s - r, so its U is what lane r-1 produced one step earlier (DPP wave_shr:1),
// its UL its previous U and its L its own previous output: no LDS, no synchronisation inside a band.  Lane 63 leaves its
// row in a seam buffer for lane 0 of the next band.  All four channels move through one register (v_lerp_u8 average,
// 16-bit-lane gradient, byte-parallel add); residual words arrive as one 16-byte load per lane every 4 steps and pixels
// leave as one 16-byte store (single dwords at the two ends of a row).  RGBA only.
constexpr uint32_t RB_MAXW = 2048;  // seam buffer, pixels
// Row staging through LDS.  A lane's row moves between HBM and the wave in whole, 64-byte-aligned 16-word chunks - four
// back-to-back 16-byte requests per lane for ONE chunk of its own row, once per 16 steps - instead of one 16-byte piece every
// 4 steps: with thousands of waves x 64 rows in flight the pieces of a line used to arrive microseconds apart, L2 had dropped the
// line in between, and the kernel moved 2.2-2.4x its algorithmic bytes (profiles/r02_pmc_step_mem.json: 21.9 B/px against 8).
// Each lane owns a two-chunk ring per direction (the lane skew makes the word a lane needs at a step lane-dependent, which LDS
// addressing absorbs): residual words [0,32) + a 4-word mirror of words 0-3, so that a lane's unaligned 4-word read never
// wraps; pixels [0,32) + a 4-word spill zone behind them for the 4-word write that straddles the ring's end.
constexpr uint32_t RB_RING = 36;                       // words per lane and direction (a stride of 9 x 16 bytes: conflict-free b128)
constexpr uint32_t RB_STAGE_WORDS = 64 * RB_RING * 2;  // LDS words in front of the seam row
constexpr size_t RB_LDS_BYTES(uint32_t max_w) { return (size_t)RB_STAGE_WORDS * 4 + (size_t)max_w * 4 + 256; }
__device__ __forceinline__ uint32_t swar_add8(uint32_t a, uint32_t b) {  // per-byte a + b (mod 256)
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}
// predmode (wave-uniform): 0 = average (p2a), 1 = gradient (p3a), 2 = left (p1x), 3 = up (p1y) for interior pixels; row 0 always
// predicts from the left and column 0 from above.  Modes 2 and 3 occur in gray tiles of mode 2 only (libxpng.c:890-895).
template <int PXSZ>
__device__ __forceinline__ void recon_band_core(const TileDesc &t, uint8_t *__restrict__ dst, uint64_t bpr, const uint32_t *__restrict__ rs,
                                                uint32_t first, int predmode, uint32_t *stage, uint32_t *seam, uint32_t dbgflags) {
    const uint32_t lane = threadIdx.x & 63;
    const bool grad = predmode == 1;
    const int32_t w = (int32_t)t.w;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x4_a4r __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t *rin = stage + lane * RB_RING, *rout = stage + 64 * RB_RING + lane * RB_RING;
    for (uint32_t yb = 0; yb < t.h; yb += 64) {
        const uint32_t y = yb + lane;
        const bool active = y < t.h;
        // (a lane without a row reads the tile's last row again and stores nothing)
        const uint32_t *rsrow = rs + (uint64_t)(active ? y : t.h - 1) * t.w;
        uint8_t *drow = dst + (uint64_t)(active ? y : 0) * bpr;
        // positions: p = x + a counts words from the 64-byte boundary in front of the row's first residual word (chunk c = words
        // [16c, 16c+16), chunks 0..Lc hold the row; at most 15 words in front of / behind the row are read: the neighbouring rows
        // of the tile, or the slack around the residual plane); po = x + ao the same for the pixels of the raster row
        const int32_t a = (int32_t)(((uintptr_t)rsrow >> 2) & 15u);
        const u32x4 *rsal = reinterpret_cast<const u32x4 *>(rsrow - a);
        const int32_t Lc = (a + w - 1) >> 4;
        const int32_t ao = PXSZ == 4 ? (int32_t)(((uintptr_t)drow >> 2) & 15u) : 0;
        uint8_t *dal = drow - 4 * ao;
        const int32_t olo = active ? ao : 0, ohi = active ? ao + w : 0;  // valid pixel positions of this lane
        const uint32_t pho = (uint32_t)(ao - (int32_t)lane) & 3u;      // every 4-word pixel write of this lane starts at po = pho (mod 4)
        // RGB [r3]: the same staging in BYTE positions (a pixel is 3 bytes at any alignment): pb = 3 x + ab counts bytes from the
        // 64-byte boundary in front of the row's first byte; four pixels are one unaligned 12-byte LDS write into the lane's 128-byte
        // ring (a write that crosses its end continues in the 16-byte spill zone and is taken back when ring chunk 0 is flushed);
        // a 64-byte chunk leaves as four aligned 16-byte stores once a block (48 bytes) has completed it; only the partial
        // granules at the two ends of a row fall back to dword / byte stores.  (Round 2 stored every group as 12 bytes at its own
        // address: 64 rows x 12 bytes per wave instruction.)
        const int32_t ab = PXSZ == 3 ? (int32_t)((uintptr_t)drow & 63u) : 0;
        uint8_t *dalb = drow - ab;
        const int32_t blo = active ? ab : 0, bhi = active ? ab + 3 * w : 0;  // valid byte positions of this lane
        int32_t pb0 = ab - 3 * (int32_t)lane;                           // byte position of the block's first pixel
        int32_t flc = pb0 >> 6;                                          // next chunk to flush
        uint32_t sp_n = 0;                                               // bytes of ring chunk 0 that sit in the spill zone
        const uint32_t S = t.w + 63;
        uint32_t prev = 0, U = 0, ev = 0;
#define XPNG_RB_CLAMP(c) ((c) < 0 ? 0 : (c) > Lc ? Lc : (c))
#define XPNG_RB_LOAD(G, c) { const u32x4 *cp_ = rsal + 4 * XPNG_RB_CLAMP(c); G[0] = cp_[0]; G[1] = cp_[1]; G[2] = cp_[2]; G[3] = cp_[3]; }
#define XPNG_RB_PUT(G, c) { const uint32_t sl_ = (uint32_t)XPNG_RB_CLAMP(c) & 1u; u32x4 *rp_ = reinterpret_cast<u32x4 *>(rin + 16 * sl_); \
                            rp_[0] = G[0]; rp_[1] = G[1]; rp_[2] = G[2]; rp_[3] = G[3]; if (sl_ == 0) *reinterpret_cast<u32x4 *>(rin + 32) = G[0]; }
        int32_t p0 = a - (int32_t)lane, po0 = ao - (int32_t)lane;  // positions of this lane at the first step of the block
        u32x4 G[4];
        XPNG_RB_LOAD(G, p0 >> 4);
        XPNG_RB_PUT(G, p0 >> 4);
        XPNG_RB_LOAD(G, (p0 >> 4) + 1);
        for (uint32_t s0 = 0; s0 < S; s0 += 16, p0 += 16, po0 += 16, pb0 += 48) {
            // block of 16 steps: the chunk requested a block ago lands in the ring, the next one is requested, and the block's 16
            // residual words come out of the ring (they lie in chunks p0 >> 4 and (p0 >> 4) + 1)
            XPNG_RB_PUT(G, (p0 >> 4) + 1);
            XPNG_RB_LOAD(G, (p0 >> 4) + 2);
            u32x4 cur[4];
#pragma unroll
            for (int q = 0; q < 4; q++) cur[q] = *reinterpret_cast<const u32x4_a4r *>(rin + ((uint32_t)(p0 + 4 * q) & 31u));
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t s = s0 + 4 * q;
            if (yb > 0 && (s & 63u) == 0 && (int32_t)s < w) ev = (int32_t)(s + lane) < w ? seam[s + lane] : 0u;  // next 64 columns of the row above the band
            const uint32_t rwv[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
            uint32_t o[4];
            const int32_t x0 = (int32_t)s - (int32_t)lane;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // previous-step output of the lane above (lane 0: the seam row)
                uint32_t Unew = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)prev, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)((s + k) & 63u));
                Unew = lane == 0 ? e0 : Unew;
                const int32_t x = x0 + k;
                const bool on = active && x >= 0 && x < w;
                const uint32_t rw = rwv[k];
                const uint32_t nUL = U, nU = Unew;
                uint32_t pred;
                if (!grad) pred = __builtin_amdgcn_lerp(prev, nU, 0x01010101u);  // per byte (L + U + 1) >> 1
                else {
                    const uint32_t Le = prev & 0x00FF00FFu, Ue = nU & 0x00FF00FFu, ULe = nUL & 0x00FF00FFu;
                    const uint32_t pe = ((times3(Le + Ue) + 0x04020402u - 2u * ULe) >> 2) & 0x00FF00FFu;
                    const int lg = (int)((prev >> 8) & 255u), ug = (int)((nU >> 8) & 255u), ulg = (int)((nUL >> 8) & 255u);
                    pred = pe | (((uint32_t)(((int)times3((uint32_t)(lg + ug)) - 2 * ulg + 2) >> 2) & 255u) << 8);
                }
                pred = predmode == 2 ? prev : pred;
                pred = predmode == 3 ? nU : pred;
                pred = y == 0 ? prev : pred;   // row 0 predicts from the left,
                pred = x == 0 ? nU : pred;     // column 0 from above (libxpng.c:805-810)
                uint32_t px = (swar_add8(rw, pred) & 0x00FFFFFFu) | (rw & 0xFF000000u);  // alpha travels in the residual word
                px = (rw >> 24) ? px : 0u;     // alpha == 0: the whole pixel is 0 (libxpng.c:802)
                px = (y | (uint32_t)x) == 0 ? first : px;
                if (on) { U = nU; prev = px; }
                o[k] = px;
                if (lane == 63 && on) seam[x] = px;  // (a full band: lane 63 is an existing row)
            }
            if (PXSZ == 4) {
                // four pixels into the lane's ring (words past 31: the spill zone, taken back when chunk 0 of the ring is flushed)
                *reinterpret_cast<u32x4_a4r *>(rout + ((uint32_t)(po0 + 4 * q) & 31u)) = u32x4_a4r{o[0], o[1], o[2], o[3]};
            } else {
                // RGB: the (marker) top byte of a pixel word is dropped; four pixels are 12 consecutive bytes at any alignment
                typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
                const uint32_t q0 = o[0] & 0xFFFFFFu, q1 = o[1] & 0xFFFFFFu, q2 = o[2] & 0xFFFFFFu, q3 = o[3] & 0xFFFFFFu;
                const uint32_t ob = (uint32_t)(pb0 + 12 * q) & 127u;
                *reinterpret_cast<u32x3_a1 *>(reinterpret_cast<uint8_t *>(rout) + ob) = u32x3_a1{q0 | (q1 << 24), (q1 >> 8) | (q2 << 16), (q2 >> 16) | (q3 << 8)};
                sp_n = ob > 116u ? ob - 116u : sp_n;
            }
          }
            if (PXSZ == 3) {
#define XPNG_RB_FLUSH3()                                                                                                                \
                {                                                                                                                       \
                    const u32x4 *fp = reinterpret_cast<const u32x4 *>(rout + 16 * ((uint32_t)flc & 1u));                                \
                    u32x4 v[4] = {fp[0], fp[1], fp[2], fp[3]};                                                                          \
                    if (((uint32_t)flc & 1u) == 0) {                                                                                    \
                        const u32x4 sp = *reinterpret_cast<const u32x4 *>(rout + 32);                                                   \
                        const uint32_t spw[3] = {sp.x, sp.y, sp.z};                                                                     \
                        uint32_t vw[3] = {v[0].x, v[0].y, v[0].z};                                                                      \
                        _Pragma("unroll") for (int j = 0; j < 3; j++) {                                                                 \
                            const int32_t nb = (int32_t)sp_n - 4 * j;                                                                   \
                            const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : (1u << (8 * nb)) - 1u);                          \
                            vw[j] = (spw[j] & m) | (vw[j] & ~m);                                                                        \
                        }                                                                                                               \
                        v[0].x = vw[0]; v[0].y = vw[1]; v[0].z = vw[2];                                                                 \
                        sp_n = 0;                                                                                                       \
                    }                                                                                                                   \
                    _Pragma("unroll") for (int g = 0; g < 4; g++) {                                                                     \
                        const int32_t lo = flc * 64 + 16 * g;                                                                           \
                        if (lo >= blo && lo + 16 <= bhi) *reinterpret_cast<u32x4 *>(dalb + lo) = v[g];                                  \
                        else if (lo + 16 > blo && lo < bhi) {                                                                           \
                            const uint32_t vv[4] = {v[g].x, v[g].y, v[g].z, v[g].w};                                                    \
                            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                             \
                                const int32_t dlo = lo + 4 * k;                                                                         \
                                if (dlo >= blo && dlo + 4 <= bhi) *reinterpret_cast<uint32_t *>(dalb + dlo) = vv[k];                    \
                                else if (dlo + 4 > blo && dlo < bhi) {                                                                  \
                                    _Pragma("unroll") for (int b = 0; b < 4; b++)                                                       \
                                        if (dlo + b >= blo && dlo + b < bhi) dalb[dlo + b] = (uint8_t)(vv[k] >> (8 * b));               \
                                }                                                                                                       \
                            }                                                                                                           \
                        }                                                                                                               \
                    }                                                                                                                   \
                    flc++;                                                                                                              \
                }
                if (64 * (flc + 1) <= pb0 + 48) XPNG_RB_FLUSH3()  // (a block writes 48 bytes: it completes at most one chunk)
            }
            if (PXSZ == 4) {
                // the pixel chunk this block completed (the one holding the block's first position) leaves as four 16-byte stores
#define XPNG_RB_FLUSH(fc_)                                                                                                              \
                {                                                                                                                       \
                    const int32_t fc = (fc_);                                                                                           \
                    const u32x4 *fp = reinterpret_cast<const u32x4 *>(rout + 16 * ((uint32_t)fc & 1u));                                 \
                    u32x4 v[4] = {fp[0], fp[1], fp[2], fp[3]};                                                                          \
                    if (((uint32_t)fc & 1u) == 0) {                                                                                     \
                        const u32x4 sp = *reinterpret_cast<const u32x4 *>(rout + 32);                                                   \
                        v[0].x = pho > 0 ? sp.x : v[0].x; v[0].y = pho > 1 ? sp.y : v[0].y; v[0].z = pho > 2 ? sp.z : v[0].z;           \
                    }                                                                                                                   \
                    if (!(dbgflags & 1)) {                                                                                              \
                        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                 \
                            const int32_t pq = fc * 16 + 4 * q;                                                                         \
                            if (pq >= olo && pq + 4 <= ohi) *reinterpret_cast<u32x4 *>(dal + 4ll * pq) = v[q];                          \
                            else if (pq + 4 > olo && pq < ohi) {                                                                        \
                                const uint32_t vv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};                                                \
                                _Pragma("unroll") for (int k = 0; k < 4; k++)                                                           \
                                    if (pq + k >= olo && pq + k < ohi) *reinterpret_cast<uint32_t *>(dal + 4ll * (pq + k)) = vv[k];     \
                            }                                                                                                           \
                        }                                                                                                               \
                    }                                                                                                                   \
                }
                XPNG_RB_FLUSH(po0 >> 4)
            }
        }
        if (PXSZ == 4) XPNG_RB_FLUSH(po0 >> 4)  // (po0 was advanced past the last block: its chunk holds the tail of the row, if anything)
        if (PXSZ == 3) {
            // the tail of the row: every position up to the last block's end has been written; at most two chunks are still in the ring
            if (64 * flc < bhi) XPNG_RB_FLUSH3()
            if (64 * flc < bhi) XPNG_RB_FLUSH3()
        }
#undef XPNG_RB_FLUSH3
#undef XPNG_RB_FLUSH
#undef XPNG_RB_CLAMP
#undef XPNG_RB_LOAD
#undef XPNG_RB_PUT
    }
}

template <int PXSZ>
__global__ __launch_bounds__(64) void k_dec_recon_band(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles,
                                                       TileSel sel, const uint32_t *__restrict__ resid,
                                                       uint8_t *const *__restrict__ rasters, uint64_t bpr, uint32_t dbgflags, uint32_t j0, uint32_t j1) {
    bw_prio();
    extern __shared__ uint32_t rb_lds[];  // the lanes' staging rings, then one row of the widest tile of the launch (bottom row of the band above)
    uint32_t *seam = rb_lds + RB_STAGE_WORDS;
    const uint32_t lane = threadIdx.x & 63;
    // work items [j0, j1), one wave each; a launch with fewer workgroups than items walks them with a grid stride (recon_grid)
    for (uint32_t j = j0 + blockIdx.x; j < j1; j += gridDim.x) {
        const DecTile *d = info + j;
        const uint32_t type = d->type;
        if (type == TILE_BAD) continue;
        const TileDesc t = tiles[vtile(sel, j)];
        uint8_t *dst = rasters[t.img] + (uint64_t)t.y * bpr + (uint64_t)t.x * PXSZ;
        if (type == 0) {  // raw rows (libxpng.c:846)
            const uint8_t *src = d->blob + 4;
            const uint64_t row = (uint64_t)t.w * PXSZ;
            for (uint64_t b = lane; b < row * t.h; b += 64) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = src[b]; }
            continue;
        }
        const uint32_t kw0 = ld32u(d->blob + 8);  // first pixel from the head of k (libxpng.c:850): bytes MSB-first
        const uint32_t first = ((kw0 >> 24) & 255u) | (((kw0 >> 16) & 255u) << 8) | (((kw0 >> 8) & 255u) << 16) | (PXSZ == 4 ? (kw0 & 255u) << 24 : 0u);
        recon_band_core<PXSZ>(t, dst, bpr, resid + t.pbase, first, (int)((type >> 1) & 1), rb_lds, seam, dbgflags);
    }
}

// (Round 3 built the residual extraction INTO this kernel - a lane cutting its row's residuals out of k from per-row cursors,
// no residual plane, no k_dec_resid - and measured it: bit-exact, 13.3 instead of 15.7 ms of kernel time per step, the same
// 37-38 Gpx/s, and 66.7 instead of 46.3 B/px of HBM traffic (profiles/r03_pmc_step_mem_fold_experiment.json): 64 rows x three
// variable-rate streams per wave cannot be fetched in whole 64-byte sectors without ~32 KB of LDS rings per wave, and anything less
// re-fetches every sector once per block because L2 does not keep it for the microsecond between two blocks.  k_dec_resid reads
// the same streams in raster order, perfectly coalesced; the 4 B/px residual plane is the price of that.  Not kept)
// --------------------------------------------------------------------------------------------------
// The serial size walk of the reference decoder (libxpng.c:982: t[i].f = r.p + x; x += low24(first u32)) for a caller whose
// blobs already live in HBM: one lane per image, cnt dependent loads.  The reference trusts the sizes; here a size that is
// zero or leaves the buffer parks every later tile at the end of the buffer, where k_dec_parse rejects it (avail < 4).
__global__ void k_dec_offsets(const uint8_t *const *__restrict__ blobs, const uint64_t *__restrict__ blob_len, uint32_t cnt,
                              uint32_t nimg, uint64_t *__restrict__ off) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nimg) return;
    const uint8_t *p = blobs[b];
    const uint64_t L = blob_len[b];
    uint64_t o = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        off[(uint64_t)b * cnt + i] = o;
        const uint32_t sz = o + 4 <= L ? ld32u(p + o) & 0xFFFFFFu : 0u;
        o = sz ? (o + sz <= L ? o + sz : L) : L;
    }
}

// (re)allocate the decode workspace and bring the tile offsets to the device (tile_off == nullptr: walk them there)
inline int decode_ws_prepare(DecodeWs &ws, uint32_t B, uint64_t n_tiles, uint64_t plane, const uint64_t *tile_off, uint32_t t0,
                             uint32_t total, hipStream_t s, std::string &err, const uint8_t *const *d_blob_ptrs = nullptr,
                             const uint64_t *d_blob_len = nullptr) {
    auto bad = [&](const char *m) { err = m; return 1; };
    if (ws.cap_tiles < (uint64_t)B * n_tiles || ws.cap_plane < plane) {
        decode_ws_free(ws);
        // (the context symbol area carries one largest tile of slack: a corrupt payload can make the walk pop one queue for all
        //  of a tile's steps, i.e. read up to a tile's pixel count past that queue's start)
        const uint64_t sz[5] = {rup(plane + 8192 + 450000, 256), rup(plane + 64, 256), rup(plane + 64, 256), rup(plane + 64, 256), rup(4 * plane + 1024, 256)};
        void *pl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (ws.arena && ws.arena_bytes >= sz[0] + sz[1] + sz[2] + sz[3] + sz[4] + 256 && !probe_env("XPNG_NO_ARENA")) {
            uint8_t *q = reinterpret_cast<uint8_t *>(rup(reinterpret_cast<uintptr_t>(ws.arena), 256));
            for (int i = 0; i < 5; i++) { pl[i] = q; q += sz[i]; }
            ws.planes_in_arena = true;
        } else {
            for (int i = 0; i < 5; i++)
                if (hipMalloc(&pl[i], sz[i]) != hipSuccess) {
                    for (int j = 0; j < i; j++) (void)hipFree(pl[j]);
                    return bad("hipMalloc failed (decode workspace)");
                }
        }
        ws.d_ctxsym = (uint8_t *)pl[0]; ws.d_asym = (uint8_t *)pl[1]; ws.d_alpha = (uint8_t *)pl[2]; ws.d_nlseq = (uint8_t *)pl[3];
        ws.d_resid_alloc = (uint32_t *)pl[4];
        if (hipMalloc((void **)&ws.d_info, (uint64_t)B * n_tiles * sizeof(DecTile)) != hipSuccess || hipMalloc((void **)&ws.d_off, (uint64_t)B * n_tiles * 8) != hipSuccess ||
            hipMalloc((void **)&ws.d_wdec, (uint64_t)B * n_tiles * 10 * sizeof(WDec)) != hipSuccess ||
            hipMalloc((void **)&ws.d_dtab, (uint64_t)B * n_tiles * 10 * WD_TAB_MAX) != hipSuccess)
            return bad("hipMalloc failed (decode workspace)");
        ws.d_resid = ws.d_resid_alloc + 16;
        ws.cap_tiles = (uint64_t)B * n_tiles; ws.cap_plane = plane;
    }
    if (!tile_off) {
        if (!d_blob_ptrs || !d_blob_len) return bad("device-side size walk needs the blob tables");
        k_dec_offsets<<<(B + 63) / 64, 64, 0, s>>>(d_blob_ptrs, d_blob_len, total / B, B, ws.d_off);
        ws.last_off.clear();
        return 0;
    }
    if (ws.last_off.size() != total || ws.last_t0 != t0 || memcmp(ws.last_off.data(), tile_off, (size_t)total * 8) != 0) {
        // pageable host memory: the copy is staged synchronously, so only pay for it when the offsets changed
        if (hipMemcpyAsync(ws.d_off, tile_off, (uint64_t)total * 8, hipMemcpyHostToDevice, s) != hipSuccess) return bad("tile offset upload failed");
        if (hipStreamSynchronize(s) != hipSuccess) return bad("tile offset upload failed");
        ws.last_off.assign(tile_off, tile_off + total);
        ws.last_t0 = t0;
    }
    return 0;
}

// Launch the whole decode of tiles [t0, t1) of every image of the batch.  d_blob_ptrs / d_raster_ptrs are device arrays
// of B pointers; tile_off holds B * cnt blob offsets (image-major), relative to each image's blob buffer.
// geometry of the barrier-free reconstruction launch for tiles up to max_w x max_h (0 = use the barrier form)
inline void recon_geometry(uint32_t max_w, uint32_t max_h, uint32_t &free_ew, uint32_t &threads, uint32_t &lds) {
    const uint32_t nw = (max_h + 63) / 64;
    free_ew = 0; threads = 1024; lds = 0;
    if (probe_env("XPNG_BARRIER_RECON") || max_h > 1024 || (uint64_t)nw * max_w * 4 + 64 > 60000) return;
    free_ew = max_w; threads = nw * 64; lds = nw * max_w * 4 + 64;
}

// workgroups of a band-reconstruction launch over n work items (probe builds: XPNG_RECON_CAP)
inline uint32_t recon_grid(uint32_t n) {
    const char *e = probe_env("XPNG_RECON_CAP");
    const uint32_t cap = e ? (uint32_t)atoi(e) : 0u;
    return cap && cap < n ? cap : n;
}

inline int decode_m1_launch(DecodeWs &ws, uint32_t B, uint64_t n_tiles, uint64_t plane_total, const TileDesc *d_tiles, uint64_t W,
                            uint32_t max_w, uint32_t max_h, int pxsz, const uint8_t *const *d_blob_ptrs, const uint64_t *d_blob_len,
                            uint32_t *d_status, const uint64_t *tile_off, uint32_t t0, uint32_t t1,
                            uint8_t *const *d_raster_ptrs, hipStream_t s, std::string &err, uint64_t *dbg = nullptr,
                            const uint32_t *d_order = nullptr, uint32_t n_big = 0, uint32_t min_w = 0) {
    const uint32_t cnt = t1 - t0, total = B * cnt, spt = pxsz == 4 ? 10 : 9;
    // RGBA: the residual kernel rebuilds alpha from its symbols itself (k_dec_resid FOLD) when every tile is at least 4 pixels wide -
    // every tile of a file the reference can write (RGBA narrower than 4 px is stored at level 7); otherwise k_dec_alpha makes the plane
    const bool fold = pxsz == 4 && min_w >= 4 && !probe_env("XPNG_NO_ALPHA_FOLD");
    const TileSel sel{t0, cnt, (uint32_t)n_tiles, B, d_order};
    const uint64_t plane = plane_total;
    auto bad = [&](const char *m) { err = m; return 1; };
    dbg_count_sequence();
    if (decode_ws_prepare(ws, B, n_tiles, plane, tile_off, t0, total, s, err, d_blob_ptrs, d_blob_len)) return 1;
    // (a null workspace pointer handed to a kernel is a GPU fault, i.e. abort(): refuse to launch instead)
    if (!ws.d_info || !ws.d_off || !ws.d_wdec || !ws.d_dtab || !ws.d_ctxsym || !ws.d_asym || !ws.d_alpha || !ws.d_nlseq || !ws.d_resid || !d_tiles || !d_blob_ptrs || !d_raster_ptrs)
        return bad("internal error: a decode workspace buffer was never allocated");
    const uint64_t bpr = W * (uint64_t)pxsz;
    uint32_t free_ew, rthreads, rlds;
    recon_geometry(max_w, max_h, free_ew, rthreads, rlds);
    if (!ws.side && chain_stream_create(&ws.side) != hipSuccess) return bad("stream creation failed");
    if (!ws.ev_fork && (hipEventCreateWithFlags(&ws.ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&ws.ev_join, hipEventDisableTiming) != hipSuccess)) return bad("event creation failed");
    k_dec_parse<<<(total + 63) / 64, 64, 0, s>>>(d_blob_ptrs, ws.d_off, d_blob_len, cnt, total, spt, pxsz, d_tiles, sel, ws.d_info, d_status);
    // Many tiles in flight: instruction issue is the bound, so the rANS chains run 32 streams to a wave (rans2_wide_dec.hpp);
    // few tiles: latency is the bound and a wave per stream (scalar cursors, hot-symbol registers) is quicker.
    const bool wide = !getenv("XPNG_NARROW_RANS") && ((uint64_t)total * spt > 2048 || getenv("XPNG_WIDE_RANS"));
    const size_t pad_ch = probe_pad("XPNG_PAD_CHAIN");
    constexpr uint32_t WD_CTX_STREAMS = 32, WD_ALPHA_STREAMS = 32;  // (16 alpha streams per wave - half the LDS per workgroup, twice the waves - measures the same)
    const uint32_t groups = (total + WD_CTX_STREAMS - 1) / WD_CTX_STREAMS, agroups = (total + WD_ALPHA_STREAMS - 1) / WD_ALPHA_STREAMS;
    if (wide) if (!dbg_skip("dec_prep")) k_rans2_dec_prep<<<total * spt, 64, 0, s>>>(ws.d_info, d_tiles, sel, spt, ws.d_ctxsym, ws.d_asym, ws.d_wdec, ws.d_dtab);
    // The alpha branch (its rANS block is the longest serial chain of a tile) and the nl-context branch (nine short
    // rANS blocks, then the serial context walk) are independent until k_dec_resid: run them on two HIP streams.
    if (pxsz == 4) {
        if (hipEventRecord(ws.ev_fork, s) != hipSuccess || hipStreamWaitEvent(ws.side, ws.ev_fork, 0) != hipSuccess) return bad("fork failed");
        if (wide && !probe_env("XPNG_NARROW_ALPHA")) { if (!dbg_skip("dec_chain_a")) k_rans2_dec_chain<true, WD_ALPHA_STREAMS, true><<<agroups, 64, DecChainLds<true, WD_ALPHA_STREAMS>::BYTES + pad_ch, ws.side>>>(ws.d_info, total, 9, 1, ws.d_wdec, ws.d_dtab, ws.d_ctxsym, ws.d_asym); }
        else k_rans2_decode<15><<<total, 64, 0, ws.side>>>(ws.d_info, d_tiles, sel, 9, 1, 0, ws.d_ctxsym, ws.d_asym, dbg);
        const size_t pad_al = probe_pad("XPNG_PAD_AL");
        if (fold || dbg_skip("dec_alpha")) {} else if (wide) k_dec_alpha<256><<<total, 256, pad_al, ws.side>>>(ws.d_info, d_tiles, sel, ws.d_asym, ws.d_alpha);
        else k_dec_alpha<1024><<<total, 1024, 0, ws.side>>>(ws.d_info, d_tiles, sel, ws.d_asym, ws.d_alpha);
        if (hipEventRecord(ws.ev_join, ws.side) != hipSuccess) return bad("join record failed");
    }
    if (wide) {
        if (!dbg_skip("dec_chain_c")) k_rans2_dec_chain<false, WD_CTX_STREAMS, false><<<groups * 9, 64, DecChainLds<false, WD_CTX_STREAMS>::BYTES + pad_ch, s>>>(ws.d_info, total, 0, 9, ws.d_wdec, ws.d_dtab, ws.d_ctxsym, ws.d_asym);
        // context streams the small layout cannot hold (PROB_BITS > 12 or more than 16 symbols: never written by the reference)
        if (!dbg_skip("dec_odd")) k_rans2_decode_rest<15><<<128, 64, 0, s>>>(ws.d_info, d_tiles, sel, total, 0, 9, ws.d_ctxsym, ws.d_asym, dbg, ws.d_wdec);
        // (k_rans2_decode_rest resolves slots by binary search: 4.6 KB of LDS, so its 128 workgroups are placed at once - with the
        //  37 KB slot table of k_rans2_decode<15> they waited up to 5 ms, between the chains and the walk, to find nothing to do)
    } else {
        k_rans2_decode<12><<<total * 9, 64, 0, s>>>(ws.d_info, d_tiles, sel, 0, 9, 0, ws.d_ctxsym, ws.d_asym, dbg);
        k_rans2_decode<15><<<total * 9, 64, 0, s>>>(ws.d_info, d_tiles, sel, 0, 9, 1, ws.d_ctxsym, ws.d_asym, dbg);  // blocks with PROB_BITS > 12 only
    }
    // Two size classes: the walk's duration is the chain of the biggest tile, and the work enumeration is sorted by size, so the
    // first n_big tiles of the order (x B images) are walked beside the rest - shorter chains, the lane-per-tile form - and each
    // class goes on to its own residual extraction and reconstruction: what is left behind the longer walk is the tail of ITS tiles only.
    // [r4] On TWO streams: the small-tile class stays on the caller's stream behind the context chains; the big-tile class runs on the
    // side stream BEHIND the alpha chains (its walk, ~14 ms, starts ~13 ms later than it could and still ends before the small tiles'
    // walk, ~28 ms, does).  Rounds 2-3 gave the small-tile class a third stream: every stream alive is a hardware queue, and with
    // 4 streams per context the fifth and sixth pipeline slot LOST throughput (36 / 33 against 38.5 Gpx/s); with one side stream for
    // the encode's and the decode's alpha branch and no third stream, 64 x 6 reads 43 where 64 x 4 reads 39 (profiles/r04_experiments.txt).
    // (probe builds: XPNG_SPLIT3=1 = the three-stream form.)
    const bool band = wide && max_w <= RB_MAXW && !probe_env("XPNG_WAVEFRONT_RECON");
    const bool split = band && d_order && n_big > 0 && n_big < cnt && !probe_env("XPNG_NARROW_WALK") && !getenv("XPNG_NO_SPLIT");
    const bool split3 = split && probe_env("XPNG_SPLIT3");
    const uint32_t jb = split ? n_big * B : 0;
    const uint32_t nostore = probe_env("XPNG_DBG_NOSTORE") ? 1u : 0u;
    // Occupancy limiter of the band reconstruction: 12 KB of unused LDS per wave keep it at ~10 waves per CU.  Its scattered
    // 16-byte loads and stores (64 rows per instruction) fill the memory pipeline's queues, and the chain kernels of the other
    // pipeline slots, which touch memory once per 8-step block, then wait for their words: decode-only rate at 3 slots
    // 43 -> 51 Gpx/s, combined bench +4 % (XPNG_RECON_LDS_PAD=0 turns it off)
    const size_t dbg_pad = probe_pad("XPNG_RECON_LDS_PAD");
    const size_t pad_rs = probe_pad("XPNG_PAD_RS");
    const uint32_t lpw = probe_env("XPNG_WALK_LPW") ? (uint32_t)atoi(probe_env("XPNG_WALK_LPW")) : 64u;  // tiles per wavefront of the small-tile walk: 64, or (probe builds) 32 / 16 - the same bytes
    // ts / tb: the streams the small-tile and the big-tile tails run on
    hipStream_t ts = s, tb = s;
    if (split) {
        if (!ws.ev_ctx && (hipEventCreateWithFlags(&ws.ev_ctx, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&ws.ev_small, hipEventDisableTiming) != hipSuccess)) return bad("event creation failed");
        if (split3) {
            if (!ws.side2 && chain_stream_create(&ws.side2) != hipSuccess) return bad("stream creation failed");
            ts = ws.side2;
        } else tb = ws.side;
        hipStream_t other = split3 ? ts : tb;  // the stream that is not `s`: it starts behind the context chains
        if (hipEventRecord(ws.ev_ctx, s) != hipSuccess || hipStreamWaitEvent(other, ws.ev_ctx, 0) != hipSuccess) return bad("fork failed");
        if (!dbg_skip("walk_small")) {
            if (lpw == 32) k_dec_walk_wide<32><<<(total - jb + 31) / 32, 64, WALK_WIDE_LDS_BYTES(32) + pad_ch + probe_pad("XPNG_PAD_WALK"), ts>>>(ws.d_info, d_tiles, sel, total, ws.d_ctxsym, ws.d_nlseq, jb);
            else if (lpw == 16) k_dec_walk_wide<16><<<(total - jb + 15) / 16, 64, WALK_WIDE_LDS_BYTES(16) + pad_ch + probe_pad("XPNG_PAD_WALK"), ts>>>(ws.d_info, d_tiles, sel, total, ws.d_ctxsym, ws.d_nlseq, jb);
            else k_dec_walk_wide<64><<<(total - jb + 63) / 64, 64, WALK_WIDE_LDS_BYTES(64) + pad_ch + probe_pad("XPNG_PAD_WALK"), ts>>>(ws.d_info, d_tiles, sel, total, ws.d_ctxsym, ws.d_nlseq, jb);
        }
        // the biggest tiles' chains are the longest of the decode: they get the scalar-unit walk, one wave per tile (~40 ns per
        // step against ~95 for the lane-per-tile form; a few waves per CU, so the CU's one scalar ALU is not contended), while
        // the many smaller tiles keep the lane-per-tile form beside them (XPNG_WIDE_BIG_WALK=1: the old form for both)
        if (dbg_skip("walk_big")) {} else if (probe_env("XPNG_WIDE_BIG_WALK")) k_dec_walk_wide<64><<<(jb + 63) / 64, 64, WALK_WIDE_LDS_BYTES(64), tb>>>(ws.d_info, d_tiles, sel, jb, ws.d_ctxsym, ws.d_nlseq, 0);
        else k_dec_walk<<<jb, 64, pad_ch, tb>>>(ws.d_info, d_tiles, sel, ws.d_ctxsym, ws.d_nlseq);
        // (RGBA: the alpha symbols are complete at ev_join, recorded on the side stream right behind the alpha chains - in front of the
        //  big-tile walk when that runs there)
        if (pxsz == 4 && ts != ws.side && hipStreamWaitEvent(ts, ws.ev_join, 0) != hipSuccess) return bad("join failed");
        if (pxsz == 4) {
            if (dbg_skip("resid_small")) {} else if (fold) k_dec_resid<4, 256, true><<<total - jb, 256, pad_rs, ts>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, jb);
            else k_dec_resid<4, 256><<<total - jb, 256, pad_rs, ts>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, jb);
            if (!dbg_skip("recon_small")) k_dec_recon_band<4><<<recon_grid(total - jb), 64, RB_LDS_BYTES(max_w) + dbg_pad, ts>>>(ws.d_info, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr, nostore, jb, total);
        } else {
            k_dec_resid<3, 256><<<total - jb, 256, 0, ts>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, jb);
            k_dec_recon_band<3><<<recon_grid(total - jb), 64, RB_LDS_BYTES(max_w) + dbg_pad, ts>>>(ws.d_info, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr, 0u, jb, total);
        }
    } else if (wide && !probe_env("XPNG_NARROW_WALK")) k_dec_walk_wide<64><<<(total + 63) / 64, 64, WALK_WIDE_LDS_BYTES(64), s>>>(ws.d_info, d_tiles, sel, total, ws.d_ctxsym, ws.d_nlseq, 0);
    else k_dec_walk<<<total, 64, 0, s>>>(ws.d_info, d_tiles, sel, ws.d_ctxsym, ws.d_nlseq);
    const uint32_t nt = split ? jb : total;  // work items [0, nt): the big-tile class of a split decode (on tb), or everything (on s)
    if (pxsz == 4 && tb != ws.side && hipStreamWaitEvent(tb, ws.ev_join, 0) != hipSuccess) return bad("join failed");
    if (pxsz == 4) {
        if (dbg_skip("resid_big")) {} else if (wide && fold) k_dec_resid<4, 256, true><<<nt, 256, pad_rs, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        else if (wide) k_dec_resid<4, 256><<<nt, 256, pad_rs, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        else if (fold) k_dec_resid<4, 1024, true><<<nt, 1024, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        else k_dec_resid<4, 1024><<<nt, 1024, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        if (dbg_skip("recon_big")) {} else if (band) k_dec_recon_band<4><<<recon_grid(nt), 64, RB_LDS_BYTES(max_w) + dbg_pad, tb>>>(ws.d_info, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr, nostore, 0, nt);
        else if (free_ew) k_dec_recon<4><<<total, rthreads, rlds, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_resid, d_raster_ptrs, bpr, free_ew);
        else k_dec_recon<4><<<total, 1024, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_resid, d_raster_ptrs, bpr, 0);
    } else {
        if (wide) k_dec_resid<3, 256><<<nt, 256, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        else k_dec_resid<3, 1024><<<nt, 1024, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_asym, ws.d_nlseq, ws.d_resid, 0);
        if (band) k_dec_recon_band<3><<<recon_grid(nt), 64, RB_LDS_BYTES(max_w) + dbg_pad, tb>>>(ws.d_info, d_tiles, sel, ws.d_resid, d_raster_ptrs, bpr, 0u, 0, nt);
        else if (free_ew) k_dec_recon<3><<<total, rthreads, rlds, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_resid, d_raster_ptrs, bpr, free_ew);
        else k_dec_recon<3><<<total, 1024, 0, tb>>>(ws.d_info, d_tiles, sel, ws.d_alpha, ws.d_resid, d_raster_ptrs, bpr, 0);
    }
    // everything rejoins the caller's stream: the stream that is not `s` hands in its tail's end
    if (split) {
        hipStream_t other = split3 ? ts : tb;
        if (hipEventRecord(ws.ev_small, other) != hipSuccess || hipStreamWaitEvent(s, ws.ev_small, 0) != hipSuccess) return bad("join failed");
    }
    if (hipGetLastError() != hipSuccess) return bad("decode kernel launch failed");
    return 0;
}

}  // namespace xpng
