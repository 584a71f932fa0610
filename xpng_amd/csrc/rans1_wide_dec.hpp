// rans1_wide_dec.hpp -- throughput form of the mode-2 entropy DECODE stage (rANS v1, decompress_block, libxpng.c:262-301).
//
// The same split as rans2_wide_dec.hpp (it reuses its table layouts, WDec and helpers):
//   k_rans1_dec_prep    wave per (tile, slot): type 1 / 2 blocks are finished on the spot; for rANS blocks the decode tables
//                       are built from the frequencies k_m2_dec_parse left in `tabs`
//   k_rans1_dec_chain   lane = one state, a wave = ONE slot of 32 tiles.  v1 decodes FORWARDS: pair k gives symbols 2k (state0)
//                       and 2k+1 (state1), words are read upwards and state0 refills first (libxpng.c:295-296); all lanes start
//                       at pair 0 and a lane idles once its stream is done.  Block-synchronous like the v2 chain: a 32- or
//                       64-word LDS ring per stream topped up by loads issued one 8-step block ahead, one 16-byte symbol store
//                       per stream and block.
#pragma once
#include "common.hpp"
#include "m2_decode.hpp"
#include "rans2_wide_dec.hpp"

namespace xpng {

// slots with at most 9 symbols use the register search (small layout); the rest the LDS tables (big layout)
// big-alphabet slots of mode 2: WdLayoutA's byte layout with 2^XPNG_W1D_CBITS coarse bytes (r4 experiment: 8 instead of 9 = 772 instead
// of 1 028 bytes of LDS per resident stream, 64-slot instead of 32-slot buckets for the scan)
#ifndef XPNG_W1D_CBITS
#define XPNG_W1D_CBITS 9
#endif
struct W1dLayoutA {
    static constexpr uint32_t CBITS = XPNG_W1D_CBITS, FCN = 256, RING = 64, REGN = 9;
    static constexpr uint32_t CO_OFF = 2 * (FCN + 2), TAB = CO_OFF + (1u << CBITS);
};
__host__ __device__ inline bool w1d_small(uint32_t slot) { return m2_nominal(slot) <= WdLayout<false>::REGN; }
__device__ __constant__ const uint8_t W1D_SMALL_SLOT[11] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11};
__device__ __constant__ const uint8_t W1D_BIG_SLOT[7] = {10, 12, 13, 14, 15, 16, 17};

__global__ __launch_bounds__(64) void k_rans1_dec_prep(const M2DecTile *__restrict__ info, const TileDesc *__restrict__ tiles, TileSel sel,
                                                       const M2Blk *__restrict__ blk, const uint16_t *__restrict__ tabs,
                                                       uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                       const uint32_t *__restrict__ stream_n, WDec *__restrict__ wdec, uint8_t *__restrict__ dtab) {
    __shared__ uint32_t fc[256];
    const uint32_t j = blockIdx.x / 18, slot = blockIdx.x % 18, lane = threadIdx.x & 63;
    WDec *wd = wdec + (uint64_t)j * M2_SLOTS + slot;
    if (lane == 0) wd->kind = 0;
    const M2DecTile d = info[j];
    if (!((d.kind == 1 && slot < 17) || (d.kind == 2 && slot == 17))) return;
    const uint32_t tile = vtile(sel, j);
    const TileDesc t = tiles[tile];
    const M2Blk mb = blk[(uint64_t)tile * M2_SLOTS + slot];
    const uint64_t out_off = sbase2[tile] + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot);
    uint8_t *out = scratch2 + out_off;
    const uint32_t type = sgpr(mb.type), n = sgpr(mb.n);
    const uint32_t Nnom = m2_nominal(slot);
    const uint32_t pb = slot >= 17 ? 15 : 14;
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
    const uint16_t *F16 = tabs + ((uint64_t)tile * M2_SLOTS + slot) * 256;
    uint32_t hot0, hot1;
    {   // cum by 4-per-lane partial sums + wave scan; fc[i] = F | cum << 16
        const uint32_t b = lane * 4;
        const uint32_t f0 = b + 0 < Nnom ? F16[b + 0] : 0, f1 = b + 1 < Nnom ? F16[b + 1] : 0, f2 = b + 2 < Nnom ? F16[b + 2] : 0, f3 = b + 3 < Nnom ? F16[b + 3] : 0;
        const uint32_t tot = f0 + f1 + f2 + f3;
        uint32_t incl = tot;
        incl = wave_scan_incl(incl);
        const uint32_t c0 = incl - tot, c1 = c0 + f0, c2 = c1 + f1, c3 = c2 + f2;
        fc[b + 0] = b + 0 < Nnom ? f0 | (c0 << 16) : 0;
        fc[b + 1] = b + 1 < Nnom ? f1 | (c1 << 16) : 0;
        fc[b + 2] = b + 2 < Nnom ? f2 | (c2 << 16) : 0;
        fc[b + 3] = b + 3 < Nnom ? f3 | (c3 << 16) : 0;
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
        if ((hot1 >> 8) == 0) hot1 = hot0;
    }
    __syncthreads();
    const bool small = w1d_small(slot);
    // big-alphabet slots: 16-bit cumulative counts cum[0 .. N] (cum[N] = 2^pb, 0x8000 behind it: greater than any slot) and 2^9
    // coarse bytes - the byte layout of rans2_wide_dec.hpp's WdLayoutA, but indexed by the SLOT (residual alphabets are not
    // skewed enough for the cold-rank form to pay: measured 108 ms against 76 per 32 images).  1 KB of LDS per resident stream
    // instead of 2 KB: a chain wave holds its 32 tables for its whole life, and LDS is what the pipelined slots run out of first.
    const uint32_t co_off = small ? WdLayout<false>::CO_OFF : W1dLayoutA::CO_OFF;
    const uint32_t cbits = small ? WdLayout<false>::CBITS : W1dLayoutA::CBITS;
    uint8_t *gt = dtab + ((uint64_t)j * M2_SLOTS + slot) * WD_TAB_MAX;
    if (small) {
        uint32_t *gfc = reinterpret_cast<uint32_t *>(gt);
        for (uint32_t i = lane; i <= WdLayout<false>::FCN; i += 64) gfc[i] = i < Nnom ? fc[i] : 0xFFFFu;
    } else {
        uint16_t *gcu = reinterpret_cast<uint16_t *>(gt);
        for (uint32_t i = lane; i <= W1dLayoutA::FCN + 1; i += 64) gcu[i] = (uint16_t)(i < Nnom ? fc[i] >> 16 : (i == Nnom ? 1u << pb : 0x8000u));
    }
    {   // coarse slot -> symbol
        const uint32_t sh = pb > cbits ? pb - cbits : 0, entries = 1u << (pb - sh);
        for (uint32_t g0 = lane * 4; g0 < entries; g0 += 256) {
            uint32_t pk = 0;
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t s = (g0 + q) << sh;
                uint32_t lo = 0, hi = Nnom - 1;  // largest index with cum <= s
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if ((fc[mid] >> 16) <= s) lo = mid; else hi = mid - 1;
                }
                while (lo > 0 && (fc[lo] & 0xFFFF) == 0) lo--;  // zero-frequency symbols share their successor's cum
                pk |= lo << (8 * q);
            }
            *reinterpret_cast<uint32_t *>(gt + co_off + g0) = pk;
        }
    }
    if (lane == 0) {
        WDec w;
        w.kind = small ? 1u : 2u; w.pb = pb; w.N = Nnom; w.n = n;
        w.nw = (size - 24) >> 2;
        w.words_off = mb.cnt + 24;
        w.out_off = out_off;
        w.hot0 = hot0 & 255u; w.hot1 = hot1 & 255u;
        *wd = w;
    }
}

template <bool BIG> struct Dec1ChainLds {  // dynamic LDS layout of one launch (common.hpp: why dynamic)
    typedef typename std::conditional<BIG, W1dLayoutA, WdLayout<false>>::type L;
    static constexpr uint32_t STREAMS = 32, TSTRIDE = L::TAB + 4, RSTRIDE = 4 * L::RING + 4;
    static constexpr uint32_t OFF_RING = (STREAMS * TSTRIDE + 15u) & ~15u, OFF_OBUF = (OFF_RING + STREAMS * RSTRIDE + 31u) & ~31u;
    static constexpr size_t BYTES = OFF_OBUF + STREAMS * OB_STRIDE + 12;
};
template <bool BIG>
__global__ __launch_bounds__(64) void k_rans1_dec_chain(const M2DecTile *__restrict__ info, uint32_t total, const WDec *__restrict__ wdec,
                                                        const uint8_t *__restrict__ dtab, uint8_t *__restrict__ scratch2) {
    typedef typename std::conditional<BIG, W1dLayoutA, WdLayout<false>>::type L;
    constexpr uint32_t CBITS = L::CBITS, TAB = L::TAB, RING = L::RING, KIND = BIG ? 2 : 1, STREAMS = 32, NSLOT = BIG ? 7 : 11;
    constexpr uint32_t TSTRIDE = TAB + 4;
    constexpr uint32_t PER = RING / 8;            // words one lane requests per boundary
    constexpr uint32_t RSTRIDE = 4 * RING + 4;    // per stream: [RING words][mirror of word 0]
    typedef Dec1ChainLds<BIG> LD;
    static_assert(LD::TSTRIDE == TSTRIDE && LD::RSTRIDE == RSTRIDE && LD::STREAMS == STREAMS, "LDS layout");
    extern __shared__ __align__(32) uint8_t dec1_chain_lds[];
    uint8_t *const ltab = dec1_chain_lds;                 // [STREAMS * TSTRIDE]
    uint8_t *const ring = dec1_chain_lds + LD::OFF_RING;  // [STREAMS * RSTRIDE]
    uint8_t *const obuf = dec1_chain_lds + LD::OFF_OBUF;  // [STREAMS * OB_STRIDE]
    __builtin_amdgcn_s_setprio(XPNG_CHAIN_PRIO);
    const uint32_t lane = threadIdx.x & 63, k = lane >> 1, par = lane & 1;
    const uint32_t slot = BIG ? W1D_BIG_SLOT[blockIdx.x % NSLOT] : W1D_SMALL_SLOT[blockIdx.x % NSLOT], grp = blockIdx.x / NSLOT;
    const uint32_t j = grp * STREAMS + k;
    bool live = j < total;
    for (uint32_t ts = 0; ts < STREAMS; ts++) {
        const uint32_t jj = grp * STREAMS + ts;
        uint32_t *dst = reinterpret_cast<uint32_t *>(ltab + ts * TSTRIDE);
        if (jj >= total || sgpr(wdec[(uint64_t)jj * M2_SLOTS + slot].kind) != KIND) {
            // no chain in this slot: its lanes idle through the loop, but their table lookups must still terminate
            for (uint32_t i = lane; i < TAB / 4; i += 64) dst[i] = i < L::CO_OFF / 4 ? (BIG ? (i ? 0x80008000u : 0x80000000u) : 0xFFFFu) : 0u;
            continue;
        }
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dtab + ((uint64_t)jj * M2_SLOTS + slot) * WD_TAB_MAX);
#pragma unroll 4
        for (uint32_t i = lane; i < TAB / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const WDec *wd = wdec + (uint64_t)(live ? j : 0) * M2_SLOTS + slot;
    live = live && wd->kind == KIND;
    const uintptr_t words = live ? (uintptr_t)info[j].blob + wd->words_off : 0;
    const uintptr_t words_safe = live ? words : (uintptr_t)dtab;  // (idle lanes still issue the block loads: any readable address)
    uint8_t *out = scratch2 + (live ? wd->out_off : 0);
    const uint32_t n = live ? wd->n : 0, pb = live ? wd->pb : 14, nw = live ? wd->nw : 0;
    const uint32_t pairs = n >> 1, mask = (1u << pb) - 1, csh = pb > CBITS ? pb - CBITS : 0;
    const uint32_t ident = 1u << pb;
    const uint32_t a_fc = (uint32_t)(uintptr_t)(lds8 *)ltab + k * TSTRIDE, a_co = a_fc + L::CO_OFF;
    const uint32_t a_ring = (uint32_t)(uintptr_t)(lds8 *)ring + k * RSTRIDE;  // word slot i at a_ring + 4 i, the mirror of slot 0 at slot RING
    const uint32_t a_ob = (uint32_t)(uintptr_t)(lds8 *)obuf + k * OB_STRIDE;
    auto ring_w = [&](uint32_t idx) __attribute__((always_inline)) -> lds32 * { return (lds32 *)(uintptr_t)(a_ring + 4 * (idx & (RING - 1))); };
    // ---- initial ring contents: the first RING words; the states sit right below the words
    uint32_t rw = 0;                          // next word to read is words[rw]
    uint32_t hi = nw < RING ? nw : RING;      // words [0, hi) are resident
    for (uint32_t q = 0; q < RING / 2; q++) {
        const uint32_t a = 2 * q + par;
        if (a < hi) *ring_w(a) = gld32u(words + 4ull * a);
    }
    *(lds32 *)(uintptr_t)(a_ring + 4 * RING) = *ring_w(0);
    uint32_t slo = 0x80000000u, shi = 0;
    if (live) { const uintptr_t sp = words - 16 + 8 * par; slo = gld32u(sp); shi = gld32u(sp + 4); }
    uint32_t hif = hi;                        // highest word index resident or in flight (exclusive)
    uint32_t fa0 = 0, fhi = 0, fsh = 0;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    uint32_t qx = 0;
    uint32_t wi = 0;                          // ring slot of w1 = words[rw]; w2 = words[rw + 1] sits 4 bytes above (mirror after the last slot)
    uint32_t w1, w2;
    auto fetch_w = [&]() __attribute__((always_inline)) {
        const lds32 *p = (const lds32 *)(uintptr_t)(a_ring + 4 * wi);
        w1 = p[0]; w2 = p[1];
    };
    fetch_w();
    uint32_t cm[10];
#pragma unroll
    for (int i = 1; i <= 9; i++) {
        cm[i] = 1u << pb;
        if (!BIG && live && (uint32_t)i < wd->N) cm[i] = *(const lds32 *)(uintptr_t)(a_fc + 4 * i) >> 16;
    }
    auto step = [&](bool act, bool refill, uint32_t obpos) __attribute__((always_inline)) -> uint32_t {
        const uint32_t slot_ = slo & mask;
        uint32_t sym, F, off;
        if (!BIG) {
            // (selects by value through pick32: see k_rans2_dec_chain)
            const uint32_t c1 = cm[1], c2 = cm[2], c3 = cm[3], c4 = cm[4], c5 = cm[5], c6 = cm[6], c7 = cm[7], c8 = cm[8], c9 = cm[9];
            const bool b2 = slot_ >= c4;
            const bool b1 = slot_ >= pick32(b2, c6, c2);
            const uint32_t t3 = pick32(b2, pick32(b1, c7, c5), pick32(b1, c3, c1));
            const bool b0 = slot_ >= t3;
            const bool b3 = b2 && b1 && b0 && slot_ >= c8;
            sym = (b2 ? 4u : 0u) + (b1 ? 2u : 0u) + (b0 ? 1u : 0u) + (b3 ? 1u : 0u);
            const uint32_t lo01 = pick32(b0, c1, 0u), lo23 = pick32(b0, c3, c2), lo45 = pick32(b0, c5, c4), lo67 = pick32(b0, c7, c6);
            const uint32_t hi01 = pick32(b0, c2, c1), hi23 = pick32(b0, c4, c3), hi45 = pick32(b0, c6, c5), hi67 = pick32(b0, c8, c7);
            const uint32_t lo03 = pick32(b1, lo23, lo01), lo47 = pick32(b1, lo67, lo45), hi03 = pick32(b1, hi23, hi01), hi47 = pick32(b1, hi67, hi45);
            uint32_t cum = pick32(b2, lo47, lo03), nxt = pick32(b2, hi47, hi03);
            cum = pick32(b3, c8, cum); nxt = pick32(b3, c9, nxt);
            F = nxt - cum; off = slot_ - cum;
        } else {
            typedef uint32_t u32_a2 __attribute__((aligned(2)));
            typedef __attribute__((address_space(3))) u32_a2 lds32u;
            sym = *(const lds8 *)(uintptr_t)(a_co + (slot_ >> csh));
            uint32_t e = *(const lds32u *)(uintptr_t)(a_fc + 2 * sym);  // cum[sym] | cum[sym + 1] << 16
            while (slot_ >= (e >> 16)) { sym++; e = *(const lds32u *)(uintptr_t)(a_fc + 2 * sym); }  // cum[N] = 2^pb (and 0x8000 behind it) stops it
            // (round 3 put k_rans2_dec_chain's packed 8-way compare here - one 16-byte read of the next eight cumulative counts instead
            //  of this scan - and measured 10.7 against 11.6 Gpx/s on the level-2 leg: the class streams' 32-slot buckets hold one or
            //  two boundaries, so the scan is short and the compare's fixed cost - a third LDS round trip and ~20 VALU - loses)
            F = (e >> 16) - (e & 0xFFFFu); off = slot_ - (e & 0xFFFFu);
        }
        if (!act) { F = ident; off = slot_; }
        const uint32_t qlo = __builtin_amdgcn_alignbit(shi, slo, pb), qhi = shi >> pb;  // s >> pb
        const uint64_t r0 = (uint64_t)qlo * F + off;
        const uint32_t nlo = (uint32_t)r0, nhi = __umul24(qhi, F) + (uint32_t)(r0 >> 32);
        const bool need = refill && (nhi | (nlo >> 31)) == 0;  // s < 2^31
        const uint32_t needi = need ? 1u : 0u, other = swap_pair(needi);
        const uint32_t take = pick32(par && other, w2, w1);    // state0 refills first (libxpng.c:295-296)
        shi = need ? nlo : nhi;
        slo = need ? take : nlo;
        const uint32_t cons = needi + other;
        wi = (wi + cons) & (RING - 1);
        fetch_w();
        *(lds8 *)(uintptr_t)(a_ob + (act ? obpos : 16u)) = (uint8_t)sym;
        return sym;
    };
    // wave trip count: pairs, plus one step for an odd tail
    uint32_t T = pairs + (n & 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(T, o); T = v > T ? v : T; }
    T = sgpr((T + 7) & ~7u);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending when the loop is entered (see k_rans2_dec_chain)
    for (uint32_t tb = 0; tb < T; tb += 8) {
        const uint32_t wi0 = wi;
        // v1 decodes FORWARDS and every lane starts at pair 0, so nothing has to be held back: a lane past the end of its stream
        // (or without one) just keeps stepping on whatever its state and its clamped ring hold - its symbols are not stored (the
        // block store below is gated), its requests are clamped at the stream's last word - and the odd tail symbol (libxpng.c:300:
        // one more from state0, no refill) is out of the state before the refill it does not need.  No predicates in the steps.
#pragma unroll
        for (int u = 0; u < 8; u++) step(true, true, 2 * (uint32_t)u + par);
        rw += (wi - wi0) & (RING - 1);
        {   // in-flight words land (unconditional, like the request below: fhi == 0 when nothing was requested)
            const uint32_t dw[9] = {q0.x, q0.y, q0.z, q0.w, PER > 4 ? q1.x : qx, q1.y, q1.z, q1.w, qx};
#pragma unroll
            for (int i = 0; i < (int)PER; i++) {
                const uint32_t a = fa0 + (uint32_t)i;
                if (a < fhi) *ring_w(a) = __builtin_amdgcn_alignbyte(dw[i + 1], dw[i], fsh);
            }
        }
        *(lds32 *)(uintptr_t)(a_ring + 4 * RING) = *ring_w(0);
        if (tb < pairs + (n & 1u) && par == 0) *reinterpret_cast<u32x4_t *>(out + 2ull * tb) = *(const lds128_al4 *)(uintptr_t)a_ob;
        hi = hif;
        {   // request the words above: keep the ring at most RING ahead of the cursor, at most 2 PER words per boundary.
            // The loads are issued on EVERY boundary, from word 0 when there is nothing to fetch (fhi = 0 then keeps the landing
            // from writing): a branch around them makes the compiler load into temporaries and copy those into the loop-carried
            // registers at once - a wait for loads it has just issued, once per block (the cure the v2 chains already had).  A
            // request near the top of the stream reads up to PER + 1 dwords from its first word, i.e. at most 36 bytes past the
            // block: inside the 64 readable bytes every blob buffer carries behind its contents (include/xpng_hip.h).
            uint32_t want = hi + 2 * PER;
            const uint32_t room = rw + RING;
            want = want < room ? want : room;
            want = want < nw ? want : nw;
            const bool any = want > hi;
            const uint32_t a0r = hi + PER * par;
            const bool req = any && a0r < want;
            const uint32_t a0 = req ? a0r : 0u;
            const uintptr_t A = words_safe + 4ull * a0;
            const gptr32 p = (gptr32)(A & ~(uintptr_t)3);
            const u32x4_a4 v0 = *(gptr128)p;
            q0 = make_uint4(v0.x, v0.y, v0.z, v0.w);
            if (PER > 4) { const u32x4_a4 v1 = *(gptr128)(p + 4); q1 = make_uint4(v1.x, v1.y, v1.z, v1.w); }
            qx = p[PER];
            fa0 = a0; fhi = req ? want : 0u; fsh = (uint32_t)(A & 3);
            hif = any ? want : hif;
        }
    }
}

inline int m2_wide_decode(DecodeWs &ws, uint32_t B, uint64_t n_tiles, uint32_t total, const M2DecTile *d_info2, const TileDesc *d_tiles,
                          TileSel sel, const M2Blk *d_blk2, const uint16_t *d_tabs2, uint8_t *d_scratch2, const uint64_t *d_sbase2,
                          const uint32_t *d_stream_n2, hipStream_t s, std::string &err) {
    const uint64_t need = (uint64_t)B * n_tiles * M2_SLOTS;
    if (ws.cap2 < need) {
        if (ws.d_wdec2) (void)hipFree(ws.d_wdec2);
        if (ws.d_dtab2) (void)hipFree(ws.d_dtab2);
        ws.d_wdec2 = nullptr; ws.d_dtab2 = nullptr; ws.cap2 = 0;
        if (hipMalloc((void **)&ws.d_wdec2, need * sizeof(WDec)) != hipSuccess || hipMalloc((void **)&ws.d_dtab2, need * WD_TAB_MAX) != hipSuccess) {
            err = "hipMalloc failed (mode-2 wide decode workspace)";
            return 1;
        }
        ws.cap2 = need;
    }
    const uint32_t groups = (total + 31) / 32;
    k_rans1_dec_prep<<<total * 18, 64, 0, s>>>(d_info2, d_tiles, sel, d_blk2, d_tabs2, d_scratch2, d_sbase2, d_stream_n2, ws.d_wdec2, ws.d_dtab2);
    // the two chain launches are independent: the big-alphabet slots run on the side stream beside the small ones
    if (!ws.side && chain_stream_create(&ws.side) != hipSuccess) { err = "stream creation failed"; return 1; }
    if (!ws.ev_fork && (hipEventCreateWithFlags(&ws.ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&ws.ev_join, hipEventDisableTiming) != hipSuccess)) { err = "event creation failed"; return 1; }
    if (hipEventRecord(ws.ev_fork, s) != hipSuccess || hipStreamWaitEvent(ws.side, ws.ev_fork, 0) != hipSuccess) { err = "fork failed"; return 1; }
    k_rans1_dec_chain<true><<<groups * 7, 64, Dec1ChainLds<true>::BYTES, ws.side>>>(d_info2, total, ws.d_wdec2, ws.d_dtab2, d_scratch2);
    if (hipEventRecord(ws.ev_join, ws.side) != hipSuccess) { err = "join record failed"; return 1; }
    k_rans1_dec_chain<false><<<groups * 11, 64, Dec1ChainLds<false>::BYTES, s>>>(d_info2, total, ws.d_wdec2, ws.d_dtab2, d_scratch2);
    if (hipStreamWaitEvent(s, ws.ev_join, 0) != hipSuccess) { err = "join failed"; return 1; }
    return 0;
}

}  // namespace xpng
