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

// Encode one block.  `in` = symbol bytes (global), n = count, nominalN = 9 or 256, pb = 12 or 15,
// out = 4-byte aligned global buffer with the capacity of common.hpp.  Returns the block size in bytes
// (uniform across the wave).  Must be called by all 64 lanes of a single-wave workgroup.
// LDS: hist[256], cum[257], tab[256].
__device__ inline uint32_t rans2_encode_block(const uint8_t *__restrict__ in, uint32_t n, uint32_t nominalN, int pb,
                                              uint8_t *__restrict__ out8, uint32_t *hist, uint32_t *cum, EncSym *tab) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *out = reinterpret_cast<uint32_t *>(out8);
    if (n == 0) {  // libxpng.c:313
        if (lane == 0) out[0] = 4;
        return 4;
    }
    // ---- histogram (the reference counts F[] while routing; same numbers)
    for (uint32_t i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&hist[in[i]], 1u);
    __syncthreads();
    // ---- alphabet: N = 1 + highest used symbol, distinct count (libxpng.c:314-317)
    uint32_t top = 0, distinct = 0;
    for (uint32_t i = lane; i < nominalN; i += 64) {
        if (hist[i]) { top = i; distinct++; }
    }
    top = ~wave_min_u32(~top);  // max
    distinct = wave_sum_u32(distinct);
    const uint32_t N = top + 1, rawBits = (uint32_t)bit_width(top);
    if (distinct == 1) {  // libxpng.c:318
        if (lane == 0) { out[0] = 8u | (1u << 24); out[1] = n | ((uint32_t)in[0] << 24); }
        return 8;
    }
    // ---- cumulative counts, scaled to 2^pb (libxpng.c:317,320).  N <= 256: serial prefix by lane 0 is
    // 256 LDS steps; do it as 4-per-lane partial sums + wave scan instead.
    {
        const uint32_t b = lane * 4;
        uint32_t h0 = b + 0 < N ? hist[b + 0] : 0, h1 = b + 1 < N ? hist[b + 1] : 0;
        uint32_t h2 = b + 2 < N ? hist[b + 2] : 0, h3 = b + 3 < N ? hist[b + 3] : 0;
        const uint32_t tot = h0 + h1 + h2 + h3;
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if ((int)lane >= d) incl += v;
        }
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

    // ---- the recurrence (libxpng.c:362-392).  Lane 0 = state0 / even symbols, lane 1 = state1 / odd.
    uint32_t *w = out + 3;
    uint64_t s = RANS_L;
    uint32_t cnt = 0;
    const uint32_t cmpl_base = 1u << pb;
    const int thr_shift = 31 - pb;
    const uint32_t steps = (n + 1) >> 1;
    // Symbols and table entries do not depend on the state, so a block of 8 steps is fetched up front
    // (8 byte loads + 8 ds_read_b128 in flight) and only the 64-bit recurrence stays serial.
    for (uint32_t k0 = 0; k0 < steps; k0 += 8) {
        EncSym e[8];
        bool valid[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t idx = 2 * (k0 + u) + lane;
            valid[u] = lane < 2 && idx < n;
            e[u] = tab[valid[u] ? in[idx] : in[0]];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t freq = e[u].freq_shift & 0xFFFF, rsh = e[u].freq_shift >> 16;
            const uint32_t emit = (valid[u] && (uint32_t)(s >> 32) >= (freq << thr_shift)) ? 1u : 0u;
            const uint32_t other = swap_pair(emit);
            if (emit) { w[cnt + (lane ? other : 0)] = (uint32_t)s; s >>= 32; }
            cnt += emit + other;
            if (valid[u]) {
                const uint64_t rcp = ((uint64_t)e[u].rcp_hi << 32) | e[u].rcp_lo;
                const uint64_t q = __umul64hi(s, rcp) >> rsh;
                s += e[u].bias + q * (uint64_t)(cmpl_base - freq);
            }
        }
    }
    cnt = __shfl(cnt, 0);
    w += cnt;
    if (lane < 2) { w[2 * lane] = (uint32_t)s; w[2 * lane + 1] = (uint32_t)(s >> 32); }  // state0, state1 (libxpng.c:394)
    w += 4;
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
    return csz;
}

}  // namespace xpng
