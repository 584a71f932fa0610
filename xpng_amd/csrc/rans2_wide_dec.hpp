// "Wide" rANS v2 block decode (decompress_block_v2, libxpng.c:429-493) for large batches of tiles.
//
// k_rans2_decode (m1_decode.hpp) gives a whole wavefront to one stream: 2 useful lanes.  With thousands of tiles in
// flight the machine is bound by instruction issue, so here the lanes are filled instead:
//
//   k_rans2_dec_prep   one wave per (tile, stream): block types 0/1/2 are finished on the spot; for rANS blocks the
//                      frequency table is parsed and the decode tables ((F | cum << 16)[256], coarse slot -> symbol
//                      [256]) are left in HBM together with a small descriptor
//   k_rans2_dec_chain  one wave = the SAME stream class (context c, or alpha) of 32 tiles: lanes 2k / 2k+1 carry the
//                      two interleaved states of tile k.  Chains of one class have similar lengths, so the wave's
//                      trip count (its longest chain) wastes little.
//
// The chain loop touches global memory only at block boundaries (every 8 steps): renormalisation words are staged
// through a 64-word LDS ring per stream, refilled by loads that are issued at one boundary and landed at the next
// (so no s_waitcnt for global memory sits inside the 8 steps), and decoded symbols leave as one aligned 16-byte
// store per stream and block.
#pragma once
#include "common.hpp"

#include <type_traits>
namespace xpng {

struct WDec {          // per (tile, stream) descriptor written by k_rans2_dec_prep
    uint32_t kind;     // 0 = nothing left to do; rANS chain to run: 1 = small table layout, 2 = big
    uint32_t pb, N, n;
    uint32_t nw;       // renormalisation words below the two states
    uint32_t words_off;  // byte offset of words[0] inside the tile blob
    uint64_t out_off;  // symbol destination, relative to ctxsym (c < 9) or asym (c == 9)
    uint32_t hot0, hot1;  // the two most probable symbols
    uint32_t rsh;         // alpha layout: the coarse bytes are indexed by (cold rank >> rsh), see k_rans2_dec_prep
};

// Decode tables of one stream: fc[FCN + 1] dwords (F | cum << 16; entries >= N hold F = 0xFFFF, cum = 0, which stops any
// scan), then the coarse slot -> symbol bytes: entry g = the symbol owning slot g << shift, the start of a short forward
// scan.  A wave runs for its slowest lane, so the scan must be short for EVERY slot: buckets are 16 slots wide.
// Two layouts, chosen per stream by k_rans2_dec_prep:
//   small (kind 1): nl-context streams as the reference writes them (pb <= 12, at most 16 symbols): 17 + 2^8 bytes/4..,
//                   32-word ring -> 16 KB of LDS per wave, so every chain of a large batch is resident at once
//   big   (kind 2): alpha streams (pb 15, up to 256 symbols); a context stream that does not fit `small` (never written
//                   by the reference encoder) is left to the one-wave-per-stream kernel
template <bool BIG> struct WdLayout {
    static constexpr uint32_t CBITS = BIG ? 10 : 8, FCN = BIG ? 256 : 16, RING = BIG ? 64 : 32;
    static constexpr uint32_t REGN = 9;  // small layout: at most 9 symbols (nl = 0..8), searched in registers
    static constexpr uint32_t CO_OFF = 4 * (FCN + 1), TAB = CO_OFF + (1u << CBITS);
};
// Alpha streams of mode 1 (pb 15, up to 256 symbols) use a denser form of the big layout: 16-bit cumulative counts
// cum[0 .. N] (cum[N] = 2^pb; entries behind it 0x8000, which stops any scan), F = cum[sym + 1] - cum[sym], and 2^9 coarse
// bytes (64-slot buckets): 1 KB instead of 2 KB of LDS per resident stream, the largest LDS holder of a pipelined batch.
struct WdLayoutA {
    static constexpr uint32_t CBITS = 9, FCN = 256, RING = 64, REGN = 9;
    static constexpr uint32_t CO_OFF = 2 * (FCN + 2), TAB = CO_OFF + (1u << CBITS);
};
constexpr uint32_t WD_TAB_MAX = WdLayout<true>::TAB;  // HBM stride of one stream's tables

// A pointer read out of a device structure is "generic" to the compiler: it would emit flat_load, which also counts on
// lgkmcnt and so couples every LDS wait to outstanding global loads.  These go through address space 1 explicitly.
typedef const __attribute__((address_space(1))) uint32_t *gptr32;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) u32x4_a4 *gptr128;
__device__ __forceinline__ uint32_t pick32(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
__device__ __forceinline__ uint32_t gld32u(uintptr_t a) {  // unaligned-safe little-endian u32 from global memory
    const gptr32 q = (gptr32)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3) * 8;
    const uint32_t lo = q[0];
    if (sh == 0) return lo;
    return (lo >> sh) | (q[1] << (32 - sh));
}

__global__ __launch_bounds__(64) void k_rans2_dec_prep(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles,
                                                       TileSel sel, uint32_t spt, uint8_t *__restrict__ ctxsym,
                                                       uint8_t *__restrict__ asym, WDec *__restrict__ wdec,
                                                       uint8_t *__restrict__ dtab) {
    bw_prio();
    __shared__ uint32_t fc[256];
    __shared__ uint32_t Fs[260];
    const uint32_t j = blockIdx.x / spt, c = blockIdx.x % spt, lane = threadIdx.x & 63;
    WDec *wd = wdec + (uint64_t)j * 10 + c;
    const DecTile d = info[j];
    if (lane == 0) wd->kind = 0;
    if (d.type == 0 || d.type == TILE_BAD) return;
    const TileDesc t = tiles[vtile(sel, j)];
    const uint8_t *in = d.blob + d.blk_off[c];
    const uint64_t out_off = c < 9 ? t.pbase + d.ctx_start[c] : t.pbase;
    uint8_t *out = (c < 9 ? ctxsym : asym) + out_off;
    const uint32_t h0 = sgpr(ld32u(in)), type = h0 >> 24;
    if (type == 0) return;
    const uint32_t csz = h0 & 0xFFFFFF;
    const uint8_t *end = in + csz;
    const uint32_t h1 = sgpr(ld32u(in + 4)), n = h1 & 0xFFFFFF, v2 = h1 >> 24;
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
    const uint32_t pb = h2 >> 24;  // 10..15, checked by k_dec_parse
    const uint8_t *words = in + 12;
    const uint8_t *table = in + 8 + 4ull * (h2 & 0xFFFFFF);
    if (lane == 0) {  // frequency table: <= 256 short fields, serial bit reader
        BitR tr{0, 0, table, end};
        for (uint32_t i = 0; i < N; i++) {
            uint32_t F;
            if (type == 3) F = tr.get(pb);
            else F = tr.get(1) ? tr.get(pb) : 0;
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
        fc[b + 0] = b + 0 < N ? f0 | (c0 << 16) : 0;
        fc[b + 1] = b + 1 < N ? f1 | (c1 << 16) : 0;
        fc[b + 2] = b + 2 < N ? f2 | (c2 << 16) : 0;
        fc[b + 3] = b + 3 < N ? f3 | (c3 << 16) : 0;
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
        if ((hot1 >> 8) == 0) hot1 = hot0;  // (a single used symbol cannot happen in a type-3/4 block; keep the entry valid anyway)
    }
    __syncthreads();
    uint8_t *gt = dtab + ((uint64_t)j * 10 + c) * WD_TAB_MAX;
    const bool small = c < 9 && pb <= 12 && N <= WdLayout<false>::REGN;
    const uint32_t co_off = small ? WdLayout<false>::CO_OFF : WdLayoutA::CO_OFF;
    const uint32_t cbits = small ? WdLayout<false>::CBITS : WdLayoutA::CBITS;
    if (small) {
        uint32_t *gfc = reinterpret_cast<uint32_t *>(gt);
        for (uint32_t i = lane; i <= WdLayout<false>::FCN; i += 64) gfc[i] = i < N ? fc[i] : 0xFFFFu;
    } else {
        uint16_t *gcu = reinterpret_cast<uint16_t *>(gt);
        // (entries behind cum[N] = 2^pb hold 0x8000: greater than any slot, and slot - entry keeps its 16-bit sign bit set)
        for (uint32_t i = lane; i <= WdLayoutA::FCN + 1; i += 64) gcu[i] = (uint16_t)(i < N ? fc[i] >> 16 : (i == N ? 1u << pb : 0x8000u));
    }
    // Alpha layout: the coarse bytes are indexed by the slot's COLD RANK (its position among the slots that belong to neither of
    // the two hot symbols, which the chain resolves in registers).  A skewed alphabet squeezes its two hundred rare symbols into
    // a few hundred slots: in slot space a 64-slot bucket of that region holds dozens of symbol boundaries, in rank space the
    // whole region usually fits the 512 entries one slot apiece (rsh = 0) and the lookup is exact.
    uint32_t rsh = 0, Ca = 0, Fa = 0, Cb = 0, Fb = 0;
    if (!small) {
        const uint32_t e0 = fc[hot0 & 255u], e1 = fc[hot1 & 255u];
        Ca = e0 >> 16; Fa = e0 & 0xFFFFu;
        if ((hot1 & 255u) != (hot0 & 255u)) { Cb = e1 >> 16; Fb = e1 & 0xFFFFu; }
        if (Fb && Cb < Ca) { const uint32_t tc = Ca, tf = Fa; Ca = Cb; Fa = Fb; Cb = tc; Fb = tf; }  // a = the lower one
        const uint32_t cold = (1u << pb) - Fa - Fb;
        while ((cold >> rsh) > (1u << cbits)) rsh++;
    }
    {   // coarse (slot or cold rank) -> symbol
        const uint32_t sh = small ? (pb > cbits ? pb - cbits : 0) : rsh, entries = small ? 1u << (pb - sh) : 1u << cbits;
        for (uint32_t g0 = lane * 4; g0 < entries; g0 += 256) {
            uint32_t pk = 0;
            for (uint32_t q = 0; q < 4; q++) {
                uint32_t s = (g0 + q) << sh;
                if (!small) {  // rank -> slot: step over the hot ranges
                    if (s >= Ca) s += Fa;
                    if (Fb && s >= Cb) s += Fb;
                    s = s < (1u << pb) ? s : (1u << pb) - 1;
                }
                uint32_t lo = 0, hi = N - 1;  // largest index with cum <= s
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if ((fc[mid] >> 16) <= s) lo = mid; else hi = mid - 1;
                }
                while (lo > 0 && (fc[lo] & 0xFFFF) == 0) lo--;  // only reachable on corrupt tables
                pk |= lo << (8 * q);
            }
            *reinterpret_cast<uint32_t *>(gt + co_off + g0) = pk;
        }
    }
    const uint8_t *sp = table - 16;  // state0 at table-16, state1 at table-8
    if (lane == 0) {
        WDec w;
        w.kind = small ? 1u : 2u; w.pb = pb; w.N = N; w.n = n;
        w.nw = (uint32_t)((sp - words) >> 2);
        w.words_off = d.blk_off[c] + 12;
        w.out_off = out_off;
        w.hot0 = hot0 & 255u; w.hot1 = hot1 & 255u;
        w.rsh = rsh;
        *wd = w;
    }
}

typedef __attribute__((address_space(3))) uint8_t lds8;
typedef __attribute__((address_space(3))) uint32_t lds32;
typedef __attribute__((address_space(3))) uint16_t lds16;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds128;
// Symbol staging of the wide DECODE chains: every stream of a wave owns OB_STRIDE bytes of LDS - the 16 symbol bytes of a block and
// a dump byte for inactive steps.  36 bytes, not 32: a step stores one byte per lane at the same position of every stream, and
// with 32 bytes per stream eight streams share each bank (an 8-deep serialised store per step; 4 conflict cycles per LDS
// instruction in the counters), with 9 words per stream the 32 streams fall on 32 banks.  The block's 16 bytes leave by a
// 4-byte aligned 16-byte read.
constexpr uint32_t OB_STRIDE = 36;
typedef u32x4_t u32x4_al4 __attribute__((aligned(4)));
typedef __attribute__((address_space(3))) u32x4_al4 lds128_al4;

// Per decode step and lane (fast path, every lane active): slot -> coarse byte -> (F | cum << 16) are two dependent LDS reads; the
// state update is one 64x32 multiply-add built from v_mad_u64_u32 + v_mad_u32_u24 (the high half of s >> pb is < 2^21,
// F < 2^16); the pair's word cursor is an LDS address that wraps inside a 256-byte aligned ring by a bit-field insert,
// and both candidate words come back with one ds_read2 (a mirror dword in front of the ring covers the wrap).
// STREAMS per wave (2 lanes each; further lanes idle).  HOT: the (F, cum) of the two most probable symbols live in registers
// and a step in which EVERY lane of the wave lands in one of them skips both table reads (which is most steps when the
// streams are skewed and the wave carries few of them: alpha runs 8 streams per wave for that reason, trading lanes for
// latency on the longest chain of the decode).
template <bool BIG, int STREAMS> struct DecChainLds {  // dynamic LDS layout of one launch (common.hpp: why dynamic)
    typedef typename std::conditional<BIG, WdLayoutA, WdLayout<false>>::type L;
    static constexpr uint32_t LTAB = BIG ? L::TAB : L::CO_OFF, TSTRIDE = LTAB + 4, RSTRIDE = 4 * L::RING + 4;
    static constexpr uint32_t OFF_RING = (STREAMS * TSTRIDE + 15u) & ~15u, OFF_OBUF = (OFF_RING + STREAMS * RSTRIDE + 31u) & ~31u;
    static constexpr size_t BYTES = OFF_OBUF + (STREAMS + 1) * OB_STRIDE + 12;
};
template <bool BIG, int STREAMS, bool HOT>
__global__ __launch_bounds__(64) void k_rans2_dec_chain(const DecTile *__restrict__ info, uint32_t total, uint32_t c_first,
                                                        uint32_t c_count, const WDec *__restrict__ wdec,
                                                        const uint8_t *__restrict__ dtab, uint8_t *__restrict__ ctxsym,
                                                        uint8_t *__restrict__ asym) {
    __builtin_amdgcn_s_setprio(XPNG_CHAIN_PRIO);  // a serial chain: its latency is the critical path, the throughput kernels beside it are not
    XPNG_PROBE_BEGIN()
    typedef typename std::conditional<BIG, WdLayoutA, WdLayout<false>>::type L;
    constexpr uint32_t CBITS = L::CBITS, TAB = L::TAB, WD_RING = L::RING, KIND = BIG ? 2 : 1;
    // LDS copy of a stream's tables: the small layout is searched in registers, so only its fc[] dwords come in (the coarse
    // bytes stay in HBM): LDS per wave is what limits how many chains of a pipelined batch are resident at once
    constexpr uint32_t LTAB = BIG ? TAB : L::CO_OFF;
    constexpr uint32_t TSTRIDE = LTAB + 4;  // LDS stride: +1 bank per table, the lanes mostly look up the same symbol
    constexpr uint32_t PER = WD_RING / 8;      // words one lane requests per boundary (the pair: a quarter of the ring)
    constexpr uint32_t RSTRIDE = 4 * WD_RING + 4;  // per stream: [mirror of the last ring word][WD_RING words]
    constexpr uint32_t WD_STREAMS = STREAMS;
    typedef DecChainLds<BIG, STREAMS> LD;
    static_assert(LD::TSTRIDE == TSTRIDE && LD::RSTRIDE == RSTRIDE, "LDS layout");
    extern __shared__ __align__(32) uint8_t dec_chain_lds[];
    uint8_t *const ltab = dec_chain_lds;                 // [WD_STREAMS * TSTRIDE]
    uint8_t *const ring = dec_chain_lds + LD::OFF_RING;  // [WD_STREAMS * RSTRIDE]
    uint8_t *const obuf = dec_chain_lds + LD::OFF_OBUF;  // [(WD_STREAMS + 1) * OB_STRIDE] per stream: 16 symbol bytes of the block + a dump byte nobody reads; the last slot belongs to the lanes without a stream (STREAMS < 32)
    const uint32_t lane = threadIdx.x & 63, kraw = lane >> 1, k = kraw < WD_STREAMS ? kraw : 0, par = lane & 1;  // (idle lanes alias stream 0's LDS harmlessly)
    const uint32_t c = c_first + blockIdx.x % c_count, grp = blockIdx.x / c_count;
    const uint32_t j = grp * WD_STREAMS + kraw;
    bool live = kraw < WD_STREAMS && j < total;
    // decode tables of the wave's streams -> LDS
    for (uint32_t ts = 0; ts < WD_STREAMS; ts++) {
        const uint32_t jj = grp * WD_STREAMS + ts;
        uint32_t *dst = reinterpret_cast<uint32_t *>(ltab + ts * TSTRIDE);
        if (jj >= total || sgpr(wdec[(uint64_t)jj * 10 + c].kind) != KIND) {
            // no chain in this slot: its lanes idle through the loop, but their table lookups must still terminate
            // (big layout: cum[0] = 0, every other entry 0x8000: symbol 0 owns every slot)
            for (uint32_t i = lane; i < LTAB / 4; i += 64) dst[i] = i < L::CO_OFF / 4 ? (BIG ? (i ? 0x80008000u : 0x80000000u) : 0xFFFFu) : 0u;  // every coarse byte names symbol 0, whose entry stops the scan
            continue;
        }
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dtab + ((uint64_t)jj * 10 + c) * WD_TAB_MAX);
#pragma unroll 4
        for (uint32_t i = lane; i < LTAB / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const WDec *wd = wdec + (uint64_t)(live ? j : 0) * 10 + c;
    live = live && wd->kind == KIND;
    const uintptr_t words = live ? (uintptr_t)info[j].blob + wd->words_off : 0;
    const uintptr_t words_safe = live ? words : (uintptr_t)dtab;  // (idle lanes still issue the block loads: any readable address)
    uint8_t *out = (c < 9 ? ctxsym : asym) + (live ? wd->out_off : 0);
    const uint32_t n = live ? wd->n : 0, pb = live ? wd->pb : 12, nw = live ? wd->nw : 0;
    const uint32_t npairs = n >> 1, mask = (1u << pb) - 1;
    const uint32_t ident = 1u << pb;  // table entry (F = 2^pb, cum = 0): the step maps s to s, which is how an idle lane waits
    // LDS byte addresses
    const uint32_t a_fc = (uint32_t)(uintptr_t)(lds8 *)ltab + k * TSTRIDE, a_co = a_fc + L::CO_OFF;
    const uint32_t a_ring = (uint32_t)(uintptr_t)(lds8 *)ring + k * RSTRIDE + 4;  // word slot i at a_ring + 4 i, the mirror at a_ring - 4
    const uint32_t a_ob = (uint32_t)(uintptr_t)(lds8 *)obuf + (kraw < WD_STREAMS ? kraw : WD_STREAMS) * OB_STRIDE;
    auto ring_w = [&](uint32_t idx) __attribute__((always_inline)) -> lds32 * { return (lds32 *)(uintptr_t)(a_ring + 4 * (idx & (WD_RING - 1))); };
    // ---- initial ring contents: the top 64 words; the states sit right above the words
    uint32_t rw = nw;                               // next word to pop is words[rw - 1]
    uint32_t lo = nw > WD_RING ? nw - WD_RING : 0;  // lowest word index resident in the ring
#pragma unroll 4
    for (uint32_t q = 0; q < WD_RING / 2; q++) {
        const uint32_t a = lo + 2 * q + par;
        if (a < nw) *ring_w(a) = gld32u(words + 4ull * a);
    }
    *(lds32 *)(uintptr_t)(a_ring - 4) = *ring_w(WD_RING - 1);
    uint32_t slo = 0, shi = 0x80000000u >> 0;  // state = shi:slo
    slo = 0x80000000u; shi = 0;
    if (live) { const uintptr_t sp = words + 4ull * nw + 8 * par; slo = gld32u(sp); shi = gld32u(sp + 4); }
    uint32_t lof = lo;                              // lowest word index resident or in flight
    uint32_t fa0 = 0, fhi = 0, fsh = 0;             // in-flight chunk: first word index, landing bound, byte misalignment
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    uint32_t qx = 0;
    uint32_t wi = (rw - 1) & (WD_RING - 1);         // ring slot of w1 = words[rw - 1]; w2 = words[rw - 2] sits 4 bytes below (mirror for slot 0)
    uint32_t w1, w2;
    auto fetch_w = [&]() __attribute__((always_inline)) {
        const lds32 *p = (const lds32 *)(uintptr_t)(a_ring - 4 + 4 * wi);
        w2 = p[0]; w1 = p[1];
    };
    fetch_w();
    uint32_t hs0 = 0, hs1 = 0, hC0 = 0, hF0 = 0x10000u, hC1 = 0, hF1 = 0;  // idle lanes: "always hit" (their step is the identity anyway)
    if (HOT && live) {
        hs0 = wd->hot0; hs1 = wd->hot1;
        if (BIG) {
            hC0 = *(const lds16 *)(uintptr_t)(a_fc + 2 * hs0); hF0 = *(const lds16 *)(uintptr_t)(a_fc + 2 * hs0 + 2) - hC0;
            hC1 = *(const lds16 *)(uintptr_t)(a_fc + 2 * hs1); hF1 = *(const lds16 *)(uintptr_t)(a_fc + 2 * hs1 + 2) - hC1;
        } else {
            const uint32_t e0 = *(const lds32 *)(uintptr_t)(a_fc + 4 * hs0), e1 = *(const lds32 *)(uintptr_t)(a_fc + 4 * hs1);
            hF0 = e0 & 0xFFFFu; hC0 = e0 >> 16; hF1 = e1 & 0xFFFFu; hC1 = e1 >> 16;
        }
    }
    // alpha layout: cold rank of a slot = slot minus the hot ranges below it; the coarse bytes are indexed by rank >> rsh
    const uint32_t rE0 = hC0 + hF0, rE1 = hC1 + hF1, rF1 = hs1 != hs0 ? hF1 : 0u;
    const uint32_t rsh = (BIG && live) ? wd->rsh : 0u;
    const bool exact = BIG && __ballot(rsh != 0) == 0;  // every stream of the wave resolves a cold slot with one table byte
    // Small layout (nl-context streams, at most 9 symbols): the cumulative counts c1..c9 live in registers and a step finds
    // its symbol by a binary search over them, instead of two dependent LDS reads and a data-dependent scan that the whole
    // wave waits for.  (c_i = 2^pb for i >= N.)
    uint32_t cm[10];
#pragma unroll
    for (int i = 1; i <= 9; i++) {
        cm[i] = 1u << pb;
        if (!BIG && live && (uint32_t)i < wd->N) cm[i] = *(const lds32 *)(uintptr_t)(a_fc + 4 * i) >> 16;
    }
    // one symbol out of this lane's state, the pair's renormalisation (state1 refills first, libxpng.c:486-487)
    auto step = [&](bool act, uint32_t obpos) __attribute__((always_inline)) -> uint32_t {
        const uint32_t slot = slo & mask;
        uint32_t sym, F, off;
        const uint32_t d0 = slot - hC0, d1 = slot - hC1;
        const bool h0 = d0 < hF0, h1 = d1 < hF1;
        if (HOT && __ballot(!(h0 || h1)) == 0) {
            sym = h0 ? hs0 : hs1; F = h0 ? hF0 : hF1; off = h0 ? d0 : d1;
        } else if (!BIG) {
            // binary search over c1..c8 in registers (three compares, a fourth for symbol 8), then [cum, next) by the same
            // decisions: ~30 VALU instructions, the same for every lane, no LDS
            // (every select goes through pick32, by value: a ?: between two array elements inside this lambda becomes a
            //  select of ADDRESSES, which sends cm[] to scratch memory and puts a scratch load on every step)
            const uint32_t c1 = cm[1], c2 = cm[2], c3 = cm[3], c4 = cm[4], c5 = cm[5], c6 = cm[6], c7 = cm[7], c8 = cm[8], c9 = cm[9];
            const bool b2 = slot >= c4;
            const bool b1 = slot >= pick32(b2, c6, c2);
            const uint32_t t3 = pick32(b2, pick32(b1, c7, c5), pick32(b1, c3, c1));
            const bool b0 = slot >= t3;
            const bool b3 = b2 && b1 && b0 && slot >= c8;
            sym = (b2 ? 4u : 0u) + (b1 ? 2u : 0u) + (b0 ? 1u : 0u) + (b3 ? 1u : 0u);
            // cum = c[sym], next = c[sym + 1] (c0 = 0)
            const uint32_t lo01 = pick32(b0, c1, 0u), lo23 = pick32(b0, c3, c2), lo45 = pick32(b0, c5, c4), lo67 = pick32(b0, c7, c6);
            const uint32_t hi01 = pick32(b0, c2, c1), hi23 = pick32(b0, c4, c3), hi45 = pick32(b0, c6, c5), hi67 = pick32(b0, c8, c7);
            const uint32_t lo03 = pick32(b1, lo23, lo01), lo47 = pick32(b1, lo67, lo45), hi03 = pick32(b1, hi23, hi01), hi47 = pick32(b1, hi67, hi45);
            uint32_t cum = pick32(b2, lo47, lo03), nxt = pick32(b2, hi47, hi03);
            cum = pick32(b3, c8, cum); nxt = pick32(b3, c9, nxt);
            F = nxt - cum; off = slot - cum;
        } else {
            // slot -> symbol without a data-dependent scan (a wave waits for its slowest lane, and a 64-slot bucket of a skewed
            // alphabet holds half a dozen narrow symbols: the serial scan cost 5-6 dependent LDS reads per cold step): the coarse
            // byte names the symbol owning the bucket's first slot, the next eight cumulative counts come in ONE 16-byte read
            // and are compared at once (packed 16-bit differences: cum is non-decreasing, so the count of "slot >= cum" is the
            // position of the first set sign bit); buckets with more than seven boundaries (rare) go round again
            typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
            typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
            typedef __attribute__((address_space(3))) u32x4_a2 lds128u;
            typedef uint32_t u32_a2 __attribute__((aligned(2)));
            typedef __attribute__((address_space(3))) u32_a2 lds32u;
            const uint32_t rank = slot - (slot >= rE0 ? hF0 : 0u) - (slot >= rE1 ? rF1 : 0u);
            const uint32_t ridx = rank >> rsh;
            sym = *(const lds8 *)(uintptr_t)(a_co + (ridx < (1u << CBITS) ? ridx : (1u << CBITS) - 1));
            const uint32_t slot2 = slot | (slot << 16);
            auto count8 = [&](uint32_t from) __attribute__((always_inline)) -> uint32_t {  // how many of cum[from + 1 .. from + 8] are <= slot
                const u32x4_a2 v = *(const lds128u *)(uintptr_t)(a_fc + 2 * from + 2);
                uint32_t m[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t vq = v[q];  // (by value: __builtin_bit_cast of the vector ELEMENT expression reads element 0)
                    const u16x2 d = __builtin_bit_cast(u16x2, slot2) - __builtin_bit_cast(u16x2, vq);
                    m[q] = __builtin_bit_cast(uint32_t, d) & 0x80008000u;  // the two sign bits
                }
                // sign bits (bit 15 / 31 of m[q]) -> bit 2q (low half: value 2q) and bit 16 + 2q (high half: value 2q + 1)
                uint32_t all = (m[0] >> 15) | (m[1] >> 13) | (m[2] >> 11) | (m[3] >> 9);
                all = (all | (all >> 15)) & 0xFFu;
                return (uint32_t)__builtin_ctz(all | 0x100u);
            };
            const bool hotlane = h0 || h1;     // (resolved in registers below: its table result is not used)
            uint32_t t = exact ? 0u : count8(sym);
            t = hotlane ? 0u : t;
            sym += t;
            if (!exact && __ballot(t == 8)) {  // (uniform) a bucket with more than seven boundaries: second round
                t = t == 8 ? count8(sym) : 0u;
                sym += t;
                if (__ballot(t == 8)) {        // still not there (dense alphabets, corrupt tables): plain scan, bounded by the sentinels
                    uint32_t c1 = *(const lds16 *)(uintptr_t)(a_fc + 2 * sym + 2);
                    while (slot >= c1 && sym < 255u) { sym++; c1 = *(const lds16 *)(uintptr_t)(a_fc + 2 * sym + 2); }
                }
            }
            sym = sym < 255u ? sym : 255u;
            const uint32_t cc = *(const lds32u *)(uintptr_t)(a_fc + 2 * sym);
            const uint32_t c0 = cc & 0xFFFFu, c1 = cc >> 16;
            F = c1 - c0; off = slot - c0;
            if (HOT) {  // lanes that landed in a hot symbol while another lane of the wave did not
                sym = h0 ? hs0 : (h1 ? hs1 : sym); F = h0 ? hF0 : (h1 ? hF1 : F); off = h0 ? d0 : (h1 ? d1 : off);
            }
        }
        if (!act) { F = ident; off = slot; }
        const uint32_t qlo = __builtin_amdgcn_alignbit(shi, slo, pb), qhi = shi >> pb;  // s >> pb
        const uint64_t r0 = (uint64_t)qlo * F + off;
        const uint32_t nlo = (uint32_t)r0, nhi = __umul24(qhi, F) + (uint32_t)(r0 >> 32);
        const bool need = (nhi | (nlo >> 31)) == 0;  // s < 2^31
        const uint32_t needi = need ? 1u : 0u, other = swap_pair(needi);
        const uint32_t take = pick32(!par && other, w2, w1);
        shi = need ? nlo : nhi;
        slo = need ? take : nlo;
        const uint32_t cons = needi + other;
        wi = (wi - cons) & (WD_RING - 1);
        fetch_w();
        *(lds8 *)(uintptr_t)(a_ob + (act ? obpos : 16u)) = (uint8_t)sym;
        return sym;
    };
    if (n & 1) {  // odd tail comes from state0 only (libxpng.c:471-476)
        const uint32_t sym = step(par == 0, (n - 1) & 15u);
        if (par == 0) out[n - 1] = (uint8_t)sym;
    }
    uint32_t T = npairs, Tmin = live ? npairs : 0xFFFFFFFFu;  // (a lane without a chain steps on the identity entry whatever the variant)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t v = __shfl_xor(T, o), v2 = __shfl_xor(Tmin, o);
        T = v > T ? v : T; Tmin = v2 < Tmin ? v2 : Tmin;
    }
    T = sgpr((T + 7) & ~7u); Tmin = sgpr(Tmin);
    // (nothing may be pending on vmcnt when the loop is entered: the compiler merges the counter state of the loop entry with
    //  that of the back edge conservatively, and a load still in flight from before the loop turns into a wait at the top of
    //  EVERY iteration - right behind the block loads the previous iteration has just issued)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (uint32_t jb = T; jb > 0;) {
        jb -= 8;
        const uint32_t wi0 = wi;
        if (jb + 8 <= Tmin) {  // every lane of the wave is inside its stream
#pragma unroll
            for (int u = 7; u >= 0; u--) step(true, 2 * (uint32_t)u + par);
        } else {
#pragma unroll
            for (int u = 7; u >= 0; u--) step(jb + (uint32_t)u < npairs, 2 * (uint32_t)u + par);
        }
        rw -= (wi0 - wi) & (WD_RING - 1);  // words the pair consumed in this block (at most two per step)
        // ---- block boundary: in-flight words land, symbols out, next words requested
        // (landing first: its wait then covers only what the previous boundary issued, 8 steps ago, not this block's store)
        XPNG_PROBE_WAIT()
        XPNG_PROBE_ISSUE_BEGIN()
        {   // (unconditional, like the request below: fhi == 0 when nothing was requested)
            const uint32_t dw[9] = {q0.x, q0.y, q0.z, q0.w, PER > 4 ? q1.x : qx, q1.y, q1.z, q1.w, qx};
#pragma unroll
            for (int i = 0; i < (int)PER; i++) {
                const uint32_t a = fa0 + (uint32_t)i;
                if (a < fhi) *ring_w(a) = __builtin_amdgcn_alignbyte(dw[i + 1], dw[i], fsh);
            }
        }
        *(lds32 *)(uintptr_t)(a_ring - 4) = *ring_w(WD_RING - 1);
        if (jb < npairs && par == 0) *reinterpret_cast<u32x4_t *>(out + 2ull * jb) = *(const lds128_al4 *)(uintptr_t)a_ob;
        lo = lof;
        {
            int32_t want = (int32_t)lo - (int32_t)(2 * PER);
            const int32_t room = (int32_t)rw - (int32_t)WD_RING;
            want = want > room ? want : room;
            want = want > 0 ? want : 0;
            {   // The loads are issued on every boundary, from a clamped address when there is nothing to fetch (fhi = 0 then keeps
                // the landing from writing): a branch around them makes the compiler load into temporaries and copy those into
                // the loop-carried registers at once - a wait for loads it has just issued, once per block.
                const bool any = (uint32_t)want < lo;
                const uint32_t a0r = (uint32_t)want + PER * par;
                const bool req = any && a0r < lo;
                const uint32_t a0 = req ? a0r : 0u;
                const uintptr_t A = words_safe + 4ull * a0;
                const gptr32 p = (gptr32)(A & ~(uintptr_t)3);
                const u32x4_a4 v0 = *(gptr128)p;
                q0 = make_uint4(v0.x, v0.y, v0.z, v0.w);
                if (PER > 4) { const u32x4_a4 v1 = *(gptr128)(p + 4); q1 = make_uint4(v1.x, v1.y, v1.z, v1.w); }
                qx = p[PER];
                fa0 = a0; fhi = req ? lo : 0u; fsh = (uint32_t)(A & 3);
                lof = any ? (uint32_t)want : lof;
            }
        }
        XPNG_PROBE_ISSUE_END()
    }
    XPNG_PROBE_END(BIG ? 4 : 3)
}

}  // namespace xpng
