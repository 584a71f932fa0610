// rans2.hpp -- rANS "v2" blocks of mode 1 (reference compress_block_v2 / decompress_block_v2,
// libxpng.c:307-493) on one wavefront per (tile, stream).
//
// The coder is a true serial recurrence per state (64-bit state, 32-bit renormalisation), so the only
// parallelism inside a block is the reference's own 2-way interleave: lane 0 carries state0 (even symbols),
// lane 1 carries state1 (odd symbols), in lock-step; each renormalisation word is placed by a 2-lane
// prefix (state0's word before state1's, exactly the reference's emission order).  Everything around the
// recurrence is wave-parallel: histogram, frequency normalisation + "steal" repair, reciprocal tables,
// raw (type 2) packing.
#pragma once
#include "common.hpp"

namespace xpng {

struct EncSym {  // 16 B: one ds_read_b128 per symbol
    uint32_t rcp_lo, rcp_hi;
    uint32_t freq_shift;  // freq | rcp_shift << 16
    uint32_t bias;
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// MSB-first bit writer used by one lane for the (short) frequency table: BITSTREAM_WRITE/FLUSH/END.
struct BitW {
    uint64_t acc;
    uint32_t pend;
    uint32_t *p;
    __device__ __forceinline__ void put(uint32_t c, uint32_t v) {
        acc = (acc << c) | v;
        pend += c;
        if (pend >= 32) { pend -= 32; *p++ = (uint32_t)(acc >> pend); }
    }
    __device__ __forceinline__ void finish() {
        if (pend > 0) { *p++ = (uint32_t)(acc << (32 - pend)); pend = 0; }
    }
};

// ---- phases shared by the v2 (mode 1) and v1 (mode 2) block encoders; all are wave-cooperative (64 lanes, one wave
// per workgroup) and communicate through LDS arrays owned by the kernel.

// symbol histogram of in[0..n) into hist[256]
// (the reference counts F[] while routing; same numbers).  16 symbols per lane per load, counted by LDS atomics into HSUB
// sub-histograms (lane % HSUB).  The streams are skewed - most alpha symbols of a tile are one value, a context stream has nine -
// and an LDS atomic serialises the lanes of an instruction that hit the same BANK, whether or not they hit the same word: with
// the sub-histograms 256 words apart (rounds 1-3: four of them) every lane counting the dominant symbol sat in one bank, 64
// deep.  Here consecutive sub-histograms are 258 words apart, i.e. two banks, so the 64 lanes spread over 16 banks, 4 deep.
constexpr uint32_t HSUB = 16, HSTRIDE = 258;
__device__ inline void rans_histogram(const uint8_t *__restrict__ in, uint32_t n, uint32_t *hist) {
    const uint32_t lane = threadIdx.x & 63;
    __shared__ uint32_t hsub[HSUB * HSTRIDE];
    for (uint32_t i = lane; i < HSUB * HSTRIDE; i += 64) hsub[i] = 0;
    __syncthreads();
    {
        uint32_t *hs = hsub + (lane % HSUB) * HSTRIDE;
        const uint32_t head = (uint32_t)((16 - ((uintptr_t)in & 15)) & 15);
        const uint32_t nh = head < n ? head : n;
        if (lane < nh) atomicAdd(&hs[in[lane]], 1u);
        const uint32_t vecs = (n - nh) >> 4;
        const uint4 *v = reinterpret_cast<const uint4 *>(in + nh);
        // (the next 16 bytes are requested before this iteration's sixteen atomics, from a clamped index: no branch around the load)
        uint4 nx = make_uint4(0, 0, 0, 0);
        if (vecs) nx = v[lane < vecs ? lane : vecs - 1];
        for (uint32_t i = lane; i < vecs; i += 64) {
            const uint4 q = nx;
            nx = v[i + 64 < vecs ? i + 64 : vecs - 1];
            const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hs[w4[k] & 255u], 1u);
                atomicAdd(&hs[(w4[k] >> 8) & 255u], 1u);
                atomicAdd(&hs[(w4[k] >> 16) & 255u], 1u);
                atomicAdd(&hs[w4[k] >> 24], 1u);
            }
        }
        for (uint32_t i = nh + (vecs << 4) + lane; i < n; i += 64) atomicAdd(&hs[in[i]], 1u);
    }
    __syncthreads();
    for (uint32_t i = lane; i < 256; i += 64) {
        uint32_t t = 0;
#pragma unroll
        for (uint32_t k = 0; k < HSUB; k++) t += hsub[k * HSTRIDE + i];
        hist[i] = t;
    }
    __syncthreads();
}

// N = 1 + highest used symbol (returned as top), number of distinct symbols (libxpng.c:314-317)
__device__ inline void rans_alphabet(const uint32_t *hist, uint32_t nominalN, uint32_t &top, uint32_t &distinct) {
    const uint32_t lane = threadIdx.x & 63;
    top = 0; distinct = 0;
    for (uint32_t i = lane; i < nominalN; i += 64) {
        if (hist[i]) { top = i; distinct++; }
    }
    top = ~wave_min_u32(~top);  // max
    distinct = wave_sum_u32(distinct);
}

// scaled cumulative counts + "steal" repair + encoder entries; on return hist[i] = normalised F[i] for i < N
__device__ inline void rans_tables(uint32_t *hist, uint32_t *cum, EncSym *tab, uint32_t N, uint32_t n, int pb) {
    const uint32_t lane = threadIdx.x & 63;
    // ---- cumulative counts, scaled to 2^pb (libxpng.c:317,320).  N <= 256: serial prefix by lane 0 is
    // 256 LDS steps; do it as 4-per-lane partial sums + wave scan instead.
    {
        const uint32_t b = lane * 4;
        uint32_t h0 = b + 0 < N ? hist[b + 0] : 0, h1 = b + 1 < N ? hist[b + 1] : 0;
        uint32_t h2 = b + 2 < N ? hist[b + 2] : 0, h3 = b + 3 < N ? hist[b + 3] : 0;
        const uint32_t tot = h0 + h1 + h2 + h3;
        uint32_t incl = tot;
        incl = wave_scan_incl(incl);
        const uint32_t ex = incl - tot;
        const uint32_t c1 = ex + h0, c2 = c1 + h1, c3 = c2 + h2, c4 = c3 + h3;  // cum[b+1..b+4]
        if (lane == 0) cum[0] = 0;
        if (b + 1 <= N) cum[b + 1] = (uint32_t)(((uint64_t)c1 << pb) / n);
        if (b + 2 <= N) cum[b + 2] = (uint32_t)(((uint64_t)c2 << pb) / n);
        if (b + 3 <= N) cum[b + 3] = (uint32_t)(((uint64_t)c3 << pb) / n);
        if (b + 4 <= N) cum[b + 4] = (uint32_t)(((uint64_t)c4 << pb) / n);
    }
    __syncthreads();
    // ---- "steal" repair, in symbol order (libxpng.c:321-328): a used symbol squeezed to width 0 takes one
    // slot from the narrowest symbol wider than 1 (first such on ties); everything between shifts by one.
    for (uint32_t i = 0; i < N; i++) {
        if (!(hist[i] && cum[i + 1] == cum[i])) continue;  // uniform: LDS values
        uint32_t best = ~0u;
        for (uint32_t j = lane; j < N; j += 64) {
            const uint32_t f = cum[j + 1] - cum[j];
            const uint32_t key = (f << 16) | j;
            if (f > 1 && key < best) best = key;
        }
        best = wave_min_u32(best);
        const uint32_t donor = best & 0xFFFF;
        __syncthreads();
        if (donor < i) { for (uint32_t j = donor + 1 + lane; j <= i; j += 64) cum[j]--; }
        else { for (uint32_t j = i + 1 + lane; j <= donor; j += 64) cum[j]++; }
        __syncthreads();
    }
    // ---- encoder table (libxpng.c:331-360); normalised F replaces the counts (libxpng.c:329)
    for (uint32_t i = lane; i < N; i += 64) {
        const uint32_t c = cum[i], F = cum[i + 1] - c;
        EncSym e;
        if (F < 2) {
            e.rcp_lo = e.rcp_hi = ~0u; e.freq_shift = F; e.bias = c + ((1u << pb) - 1);
        } else {
            uint32_t sh = 32 - (uint32_t)__clz((int)(F - 1));  // smallest sh with F <= 2^sh
            const uint64_t hi_dividend = 1ull << (sh + 31);
            const uint64_t q_hi = hi_dividend / F;
            const uint64_t lo_dividend = (uint64_t)(F - 1) + ((hi_dividend % F) << 32);
            const uint64_t q_lo = lo_dividend / F;
            const uint64_t rcp = q_lo + (q_hi << 32);
            e.rcp_lo = (uint32_t)rcp; e.rcp_hi = (uint32_t)(rcp >> 32);
            e.freq_shift = F | ((sh - 1) << 16); e.bias = c;
        }
        tab[i] = e;
    }
    __syncthreads();
    for (uint32_t i = lane; i < N; i += 64) hist[i] = cum[i + 1] - cum[i];  // hist := normalised F
    __syncthreads();

}

// Encode one block.  `in` = symbol bytes (global), n = count, nominalN = 9 or 256, pb = 12 or 15,
// out = 4-byte aligned global buffer with the capacity of common.hpp.  Returns the block size in bytes
// (uniform across the wave).  Must be called by all 64 lanes of a single-wave workgroup.
// LDS: hist[256], cum[257], tab[256].
__device__ inline uint32_t rans2_encode_block(const uint8_t *__restrict__ in, uint32_t n, uint32_t nominalN, int pb,
                                              uint8_t *__restrict__ out8, uint32_t *hist, uint32_t *cum, EncSym *tab,
                                              uint64_t *stamp = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *out = reinterpret_cast<uint32_t *>(out8);
    if (n == 0) {  // libxpng.c:313
        if (lane == 0) out[0] = 4;
        return 4;
    }
#define XPNG_STAMP(k) do { if (stamp && lane == 0) stamp[k] = __builtin_readcyclecounter(); } while (0)
    XPNG_STAMP(0);
    rans_histogram(in, n, hist);
    XPNG_STAMP(1);
    uint32_t top, distinct;
    rans_alphabet(hist, nominalN, top, distinct);
    const uint32_t N = top + 1, rawBits = (uint32_t)bit_width(top);
    if (distinct == 1) {  // libxpng.c:318
        if (lane == 0) { out[0] = 8u | (1u << 24); out[1] = n | ((uint32_t)in[0] << 24); }
        return 8;
    }
    rans_tables(hist, cum, tab, N, n, pb);
    XPNG_STAMP(2);
    // ---- the recurrence (libxpng.c:362-392).  Even lanes run state0 (even symbols), odd lanes state1 (odd
    // symbols); lanes 2..63 replicate lanes 0/1 so the loop has no per-lane validity masks (they never write).
    //  * symbols: each lane walks its own stride-2 byte sequence through a 64-bit register window (4 symbols)
    //    backed by two prefetched 8-byte chunks, so no global-load latency sits in the loop;
    //  * table entries: fetched three steps ahead of the state update (rotating registers), so the LDS latency is
    //    off the dependent chain, which is only: spill test -> 64x64 mulhi -> shift -> mad;
    //  * spilled words: positions come from a 2-bit ballot (state0's word before state1's, as the reference emits
    //    them), so the cursor is a scalar; words are staged in an LDS ring and flushed 256 at a time, coalesced.
    uint32_t *w = out + 3;
    __shared__ uint32_t ring[512];
    const uint32_t par = lane & 1;
    const uint32_t cmpl_base = 1u << pb;
    const int thr_shift = 31 - pb;
    uint64_t s = RANS_L;
    uint32_t cnt = 0, flushed = 0;  // wave-uniform
    {
        // Symbol supply: the stream is staged through a 2 KB LDS ring in 1 KB units, copied coalesced by the whole wave (the ring
        // keeps the stream's 16-byte phase in memory; its first 16 bytes are mirrored behind its end, so that an 8-byte window
        // read at any byte position never wraps); the unit after the newest one sits prefetched in registers.
        // The recurrence itself runs in straight-line double blocks of 8 steps in which NOTHING waits on a load just issued:
        //   * a block's 8 symbols (4 per state) are one unaligned 8-byte LDS read issued two blocks ahead;
        //   * its four table entries (16 B each) are read one block ahead, with addresses cut out of that window;
        //   * the renormalisation (spill test -> ring store -> shift) sits behind a wave-uniform branch that skewed streams
        //     take once in tens of steps, so the common step is: decode entry, compare, 64x64 high multiply, shift, mad;
        //   * ring flush and unit refill are checked once per 128 steps, outside the straight-line loop.
        __shared__ __align__(16) uint8_t sring[2048 + 16];
        const uint8_t *inA = reinterpret_cast<const uint8_t *>((uintptr_t)in & ~(uintptr_t)15);
        const uint32_t p0 = (uint32_t)((uintptr_t)in & 15);  // ring position of symbol 0
        const uint4 *src = reinterpret_cast<const uint4 *>(inA) + lane;
        {
            const uint4 u0 = src[0];
            reinterpret_cast<uint4 *>(sring)[lane] = u0;
            if (lane == 0) reinterpret_cast<uint4 *>(sring)[128] = u0;  // mirror
            reinterpret_cast<uint4 *>(sring)[64 + lane] = src[64];
        }
        uint4 pre = src[128];
        uint32_t filled = 2;        // units already in the ring (uniform)
        __syncthreads();
        typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
        auto window = [&](uint32_t pos) -> u32x2_a1 { return *reinterpret_cast<const u32x2_a1 *>(sring + (pos & 2047u)); };
        // entry of the symbol in byte (2u + par) of a window
        const uint32_t bsh = 8u * par;
        auto entry = [&](const u32x2_a1 &W, int u) -> EncSym {
            const uint32_t d = u < 2 ? W.x : W.y;
            const uint32_t sy = (d >> (bsh + 16u * (uint32_t)(u & 1))) & 255u;
            return tab[sy];
        };
        auto step = [&](const EncSym &e) __attribute__((always_inline)) {
            const uint32_t freq = e.freq_shift & 0xFFFF, rsh = e.freq_shift >> 16;
            const bool emit = (uint32_t)(s >> 32) >= (freq << thr_shift);
            const uint32_t m = sgpr((uint32_t)__ballot(emit) & 3u);  // bit0: state0 spills, bit1: state1 spills
            if (m) {  // uniform, rare for skewed streams
                const uint32_t e0 = m & 1u;
                if (emit) {
                    if (lane < 2) ring[(cnt + (par ? e0 : 0u)) & 511u] = (uint32_t)s;  // state0's word first (libxpng.c:370-373)
                    s >>= 32;
                }
                cnt += e0 + (m >> 1);
            }
            const uint64_t rcp = ((uint64_t)e.rcp_hi << 32) | e.rcp_lo;
            const uint64_t q = __umul64hi(s, rcp) >> rsh;
            s += e.bias + q * (uint64_t)(cmpl_base - freq);
        };
        const uint32_t pairs = sgpr(n >> 1);  // steps in which both states code a symbol
        const uint32_t n8 = pairs >> 3;        // straight-line double blocks
        uint32_t pos = sgpr(p0);               // ring position of the next block's first symbol (uniform)
        uint32_t db = 0;
        if (n8) {
            u32x2_a1 Wa = window(pos), Wb = window(pos + 8);
            EncSym Ea[4], Eb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) Ea[u] = entry(Wa, u);
            while (db < n8) {
                // ---- maintenance, once per 16 double blocks (128 steps: at most 256 words, exactly 256 symbol bytes)
                if (cnt - flushed >= 256) {
                    __syncthreads();
                    for (uint32_t i = lane; i < 256; i += 64) w[flushed + i] = ring[(flushed + i) & 511u];
                    flushed += 256;
                    __syncthreads();
                }
                if (filled * 1024u < pos + 256u + 48u) {  // the windows of this round reach into the next unit: rotate it in
                    __syncthreads();
                    reinterpret_cast<uint4 *>(sring)[(filled & 1u) * 64 + lane] = pre;
                    if ((filled & 1u) == 0 && lane == 0) reinterpret_cast<uint4 *>(sring)[128] = pre;
                    filled++;
                    pre = src[filled * 64];
                    __syncthreads();
                }
                const uint32_t stop = sgpr(db + 16 < n8 ? db + 16 : n8);
#pragma unroll 1
                for (; db < stop; db++) {
                    // block A: its entries Ea are here; window of block B is here (Wb); request the window after, B's entries
                    Wa = window(pos + 16);
#pragma unroll
                    for (int u = 0; u < 4; u++) Eb[u] = entry(Wb, u);
#pragma unroll
                    for (int u = 0; u < 4; u++) step(Ea[u]);
                    // block B
                    Wb = window(pos + 24);
#pragma unroll
                    for (int u = 0; u < 4; u++) Ea[u] = entry(Wa, u);
#pragma unroll
                    for (int u = 0; u < 4; u++) step(Eb[u]);
                    pos += 16;
                }
            }
        }
        // ---- tail: fewer than 8 pair steps and the odd symbol, one at a time straight from the ring
        __syncthreads();
        if (cnt - flushed >= 256) {  // (the last round may have left up to 511 words staged: make room for the tail's)
            for (uint32_t i = lane; i < 256; i += 64) w[flushed + i] = ring[(flushed + i) & 511u];
            flushed += 256;
            __syncthreads();
        }
        if (filled * 1024u < pos + 64u) {
            reinterpret_cast<uint4 *>(sring)[(filled & 1u) * 64 + lane] = pre;
            if ((filled & 1u) == 0 && lane == 0) reinterpret_cast<uint4 *>(sring)[128] = pre;
            filled++;
            __syncthreads();
        }
        for (uint32_t k = n8 * 8; k < pairs; k++) {
            step(tab[sring[(pos + par) & 2047u]]);
            pos += 2;
        }
        if (n & 1) {  // odd tail: state0 only (libxpng.c:382-392); odd lanes skip the update
            const EncSym e = tab[sring[pos & 2047u]];
            const uint32_t freq = e.freq_shift & 0xFFFF, rsh = e.freq_shift >> 16;
            const bool emit = par == 0 && (uint32_t)(s >> 32) >= (freq << thr_shift);
            const uint32_t e0 = sgpr((uint32_t)__ballot(emit) & 1u);
            if (emit) { if (lane == 0) ring[cnt & 511u] = (uint32_t)s; s >>= 32; }
            cnt += e0;
            if (par == 0) {
                const uint64_t rcp = ((uint64_t)e.rcp_hi << 32) | e.rcp_lo;
                const uint64_t q = __umul64hi(s, rcp) >> rsh;
                s += e.bias + q * (uint64_t)(cmpl_base - freq);
            }
        }
        __syncthreads();
        for (uint32_t i = flushed + lane; i < cnt; i += 64) w[i] = ring[i & 511u];
    }
    w += cnt;
    if (lane < 2) { w[2 * lane] = (uint32_t)s; w[2 * lane + 1] = (uint32_t)(s >> 32); }  // state0, state1 (libxpng.c:394)
    w += 4;
    __syncthreads();
    XPNG_STAMP(3);
    // ---- header + frequency table (libxpng.c:396-415)
    const uint32_t sparseBits = N + distinct * (uint32_t)pb;
    const bool sparse = sparseBits < N * (uint32_t)pb;
    uint32_t csz = 0;
    if (lane == 0) {
        out[1] = n | ((N - 2) << 24);
        out[2] = (uint32_t)(w - (out + 2)) | ((uint32_t)pb << 24);
        BitW tb{0, 0, w};
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t F = hist[k];
            if (!sparse) tb.put((uint32_t)pb, F);
            else if (F) tb.put((uint32_t)pb + 1, F + (1u << pb));
            else tb.put(1, 0);
        }
        tb.finish();
        csz = (uint32_t)((uint8_t *)tb.p - out8);
        out[0] = csz | ((3u + (sparse ? 1u : 0u)) << 24);
    }
    csz = __shfl(csz, 0);
    XPNG_STAMP(4);
    // ---- raw fallback, type 2 (libxpng.c:417-424): rawBits per symbol, MSB first, parallel over words
    const uint64_t rawTotalBits = (uint64_t)rawBits * n;
    const uint32_t rawWords = (uint32_t)((rawTotalBits + 31) >> 5);
    if (csz >= 8 + 4 * rawWords) {
        __syncthreads();
        for (uint32_t wi = lane; wi < rawWords; wi += 64) {
            const uint64_t b0 = (uint64_t)wi * 32;
            uint32_t j = (uint32_t)(b0 / rawBits);
            uint32_t word = 0;
            for (; j < n; j++) {
                const int64_t rel = (int64_t)((uint64_t)j * rawBits) - (int64_t)b0;  // first bit of symbol j inside the word
                if (rel >= 32) break;
                const int sh = 32 - (int)rel - (int)rawBits;
                const uint32_t v = in[j];
                word |= sh >= 0 ? (sh < 32 ? v << sh : 0u) : v >> (-sh);
            }
            out[2 + wi] = word;
        }
        csz = 8 + 4 * rawWords;
        if (lane == 0) { out[0] = csz | (2u << 24); out[1] = n | (rawBits << 24); }
    }
    XPNG_STAMP(5);
    return csz;
}

}  // namespace xpng
