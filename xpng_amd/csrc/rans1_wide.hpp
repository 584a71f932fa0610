// rans1_wide.hpp -- throughput form of the mode-2 entropy ENCODE stage (rANS v1, compress_block, libxpng.c:160-260).
//
// k_rans1_encode (m2_encode.hpp) gives a wavefront to one (tile, stream) pair: two live lanes.  With a batch in flight the
// same block bytes come from three launches, as for mode 1 (rans2_wide.hpp):
//
//   k_rans1_prep    wave per (tile, slot): histogram -> alphabet -> tables; empty and one-symbol blocks are finished here
//   k_rans1_chain   lane = one rANS state; a wave carries ONE slot (stream) of 16 or 32 tiles.  v1 runs BACKWARDS over the
//                   symbols (the decoder runs forwards) and, when both states spill in a step, state1's word comes first.
//                   Block-synchronous: a lane's 8 symbols of a block are one aligned 16-byte load issued a block earlier;
//                   words are staged in LDS in emission order and leave as 16-byte stores at the block boundary
//   k_rans1_finish  wave per (tile, slot): states, block type, the frequency-table "piece" for the shared bit stream b
#pragma once
#include "common.hpp"
#include <type_traits>
#include "m2_encode.hpp"
#include "rans2.hpp"

namespace xpng {

struct W1Prep {
    uint32_t kind;  // 1 = needs chain + finish; 0 = nothing to do (block final, or the slot is not used by this tile)
    uint32_t N, distinct, cnt;
    uint64_t st[2];
};
constexpr uint32_t W1_TAB_BYTES = 4096;  // HBM stride of one (tile, slot) encoder table (256 x 16 B)
__host__ __device__ inline bool w1_big_slot(uint32_t slot) { return slot >= 15; }  // alphabets of 128 / 256 symbols
constexpr uint32_t W1_SMALL_SLOTS = 15, W1_BIG_SLOTS = 6;                           // slots 0..14 and 15..20

__global__ __launch_bounds__(64) void k_rans1_prep(const TileDesc *__restrict__ tiles, TileSel sel, const uint32_t *__restrict__ flags,
                                                   uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                   const uint32_t *__restrict__ stream_n, M2Blk *__restrict__ blk,
                                                   W1Prep *__restrict__ prep, uint8_t *__restrict__ wtab, uint16_t *__restrict__ wF) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cum[260];
    __shared__ EncSym tab[256];
    const uint32_t tile = vtile(sel, blockIdx.x / M2_SLOTS), slot = blockIdx.x % M2_SLOTS, lane = threadIdx.x & 63;
    W1Prep *p = prep + (uint64_t)tile * M2_SLOTS + slot;
    if (lane == 0) p->kind = 0;
    const TileDesc t = tiles[tile];
    const uint32_t f = flags[tile];
    const bool single = !(f & M2F_NOT_SINGLE), gray = !single && !(f & M2F_NOT_GRAY);
    if (single || (gray != (slot >= 17))) return;  // colour tiles run slots 0..16, gray tiles 17..20
    uint8_t *sc = scratch2 + sbase2[tile];
    const uint8_t *in = sc + m2_off_stream(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot);
    const uint32_t n = sgpr(stream_n[(uint64_t)tile * M2_SLOTS + slot]);
    const uint32_t Nnom = m2_nominal(slot);
    const int pb = slot >= 17 ? 15 : 14;
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
    EncSym *gt = reinterpret_cast<EncSym *>(wtab + ((uint64_t)tile * M2_SLOTS + slot) * W1_TAB_BYTES);
    uint16_t *gF = wF + ((uint64_t)tile * M2_SLOTS + slot) * 256;
    for (uint32_t i = lane; i < N; i += 64) { gt[i] = tab[i]; gF[i] = (uint16_t)hist[i]; }
    if (lane == 0) *p = W1Prep{1, N, distinct, 0, {0, 0}};
}

// BIG: slots 15..20 (tables of 2-4 KB), 16 tiles per wave (32 lanes); else slots 0..14 (tables <= 1 KB), 32 tiles per wave.
template <bool BIG> constexpr uint32_t rans1_chain_ltab_bytes() { return (BIG ? 16u : 32u) * ((BIG ? 4096u : 1024u) + 16u); }
template <bool BIG> constexpr size_t rans1_chain_lds_bytes() { return rans1_chain_ltab_bytes<BIG>() + (BIG ? 16u : 32u) * WB_STRIDE * 4u; }  // dynamic LDS (common.hpp: why dynamic)
template <bool BIG>
__global__ __launch_bounds__(64) void k_rans1_chain(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t total,
                                                    uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                    const uint32_t *__restrict__ stream_n, W1Prep *__restrict__ prep,
                                                    const uint8_t *__restrict__ wtab) {
    constexpr uint32_t TPW = BIG ? 16 : 32, TAB = BIG ? 4096 : 1024, TSTRIDE = TAB + 16, NSLOT = BIG ? W1_BIG_SLOTS : W1_SMALL_SLOTS;
    static_assert(TPW * TSTRIDE == rans1_chain_ltab_bytes<BIG>() && (TPW * TSTRIDE) % 16 == 0, "LDS layout");
    extern __shared__ __align__(16) uint8_t rans1_chain_lds[];
    uint8_t *const ltab = rans1_chain_lds;                                                  // [TPW * TSTRIDE]
    uint32_t *const wbuf = reinterpret_cast<uint32_t *>(rans1_chain_lds + TPW * TSTRIDE);   // [TPW * WB_STRIDE] per stream: 16 staged words + dump words nobody reads (common.hpp)
    __builtin_amdgcn_s_setprio(XPNG_CHAIN_PRIO);
    const uint32_t lane = threadIdx.x & 63, k = lane >> 1, par = lane & 1;
    const uint32_t slot = (BIG ? 15u : 0u) + blockIdx.x % NSLOT, grp = blockIdx.x / NSLOT;
    const uint32_t j = grp * TPW + k;
    bool live = k < TPW && j < total;
    const int pb = slot >= 17 ? 15 : 14;
    for (uint32_t ts = 0; ts < TPW; ts++) {
        const uint32_t jj = grp * TPW + ts;
        if (jj >= total) break;
        const uint64_t rec = (uint64_t)vtile(sel, jj) * M2_SLOTS + slot;
        if (sgpr(prep[rec].kind) != 1) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(wtab + rec * W1_TAB_BYTES);
        uint4 *dst = reinterpret_cast<uint4 *>(ltab + ts * TSTRIDE);
        for (uint32_t i = lane; i < TAB / 16; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t tile = vtile(sel, live ? j : 0);
    const TileDesc *t = tiles + tile;
    uint8_t *sc = scratch2 + sbase2[tile];
    W1Prep *p = prep + (uint64_t)tile * M2_SLOTS + slot;
    live = live && p->kind == 1;
    const uint8_t *in = sc + m2_off_stream(t->n, stream_n + (uint64_t)tile * M2_SLOTS, slot);  // 64-byte aligned
    const uint32_t n = live ? stream_n[(uint64_t)tile * M2_SLOTS + slot] : 0;
    const uint32_t pairs = n >> 1;
    // chunk c = pairs 8c .. 8c+7 (16 bytes); a lane walks its chunks from the top one down to 0, and all lanes of the wave reach
    // chunk 0 together: wave chunk counter C runs Cmax-1 .. 0 and a lane is inside its stream when C < nchunks
    const uint32_t nchunks = (pairs + 7) >> 3;
    uint32_t Cmax = nchunks, Cfull = live ? pairs >> 3 : 0u;  // chunks below Cfull are complete (8 pairs) for this lane
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t v = __shfl_xor(Cmax, o), v2 = __shfl_xor(Cfull, o);
        Cmax = v > Cmax ? v : Cmax; Cfull = v2 < Cfull ? v2 : Cfull;
    }
    Cmax = sgpr(Cmax); Cfull = sgpr(Cfull);
    const EncSym *tab = reinterpret_cast<const EncSym *>(ltab + (k < TPW ? k : 0) * TSTRIDE);
    uint32_t *out = reinterpret_cast<uint32_t *>(sc + m2_off_blk(t->n, stream_n + (uint64_t)tile * M2_SLOTS, slot));
    uint32_t *w = out + 4;
    uint32_t *wb = wbuf + (k < TPW ? k : 0) * WB_STRIDE;
    const uint32_t dumpw = wb_dump(k < TPW ? k : 0, par);
    const uint32_t cmpl_base = 1u << pb;
    const int thr_shift = 31 - pb;
    uint64_t s = RANS_L;
    auto put = [&](const EncSym &e) __attribute__((always_inline)) {
        const uint32_t freq = e.freq_shift & 0xFFFF, rsh = e.freq_shift >> 16;
        const uint64_t rcp = ((uint64_t)e.rcp_hi << 32) | e.rcp_lo;
        const uint64_t q = __umul64hi(s, rcp) >> rsh;
        s += e.bias + q * (uint64_t)(cmpl_base - freq);
    };
    if (live && (n & 1) && !par) put(tab[in[n - 1]]);  // libxpng.c:218-225: the odd tail goes to state0, no spill test
    uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;           // in flight: the chunk below the current one
    // (unconditional, on a clamped chunk index; the landing below sits in front of the chunk's stores; the loop is entered with
    //  nothing pending: see k_rans2_chain2 / DESIGN.md for what each of these avoids)
    auto request = [&](int32_t c) __attribute__((always_inline)) {
        const uint32_t cc = c < 0 ? 0u : ((uint32_t)c < nchunks ? (uint32_t)c : (nchunks ? nchunks - 1 : 0u));
        const uint4 v = *reinterpret_cast<const uint4 *>(in + 16ull * cc);
        f0 = v.x; f1 = v.y; f2 = v.z; f3 = v.w;
    };
    uint32_t sy0 = 0, sy1 = 0;  // this lane's 8 symbols of the current chunk, one per byte (pair 8c + i in byte i)
    auto land = [&]() __attribute__((always_inline)) {
        const uint32_t selb = par ? 0x07050301u : 0x06040200u;  // odd / even bytes of a dword pair
        sy0 = __builtin_amdgcn_perm(f1, f0, selb);
        sy1 = __builtin_amdgcn_perm(f3, f2, selb);
    };
    request((int32_t)Cmax - 1); land(); request((int32_t)Cmax - 2);
    uint32_t cnt = 0;
    for (uint32_t C = Cmax; C > 0;) {
        C--;
        uint32_t cb = 0;  // words the pair has staged in this chunk
        // the chunk's eight table entries are read up front, back to back (inside the steps each 16-byte LDS read sat in front of
        // the spill test that needs its frequency: one LDS latency per step instead of one per chunk)
        EncSym E[8];
#pragma unroll
        for (int u = 0; u < 8; u++) E[u] = tab[((u < 4 ? sy0 : sy1) >> (8 * (u & 3))) & 255u];
        auto steps8 = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;  // every lane of the wave has all 8 pairs of this chunk: no activity test
#pragma unroll
            for (int u = 7; u >= 0; u--) {
                const EncSym e = E[u];
                const bool act = FULL || 8 * C + (uint32_t)u < pairs;
                const uint32_t freq = e.freq_shift & 0xFFFF;
                const uint32_t emit = (act && (uint32_t)(s >> 32) >= (freq << thr_shift)) ? 1u : 0u;
                const uint32_t other = swap_pair(emit);
                wb[emit ? cb + (par ? 0u : other) : dumpw] = (uint32_t)s;  // state1's word first (libxpng.c:229-236)
                if (emit) s >>= 32;
                cb += emit + other;
                if (act) put(e);
            }
        };
        if (C < Cfull) steps8(std::true_type{});
        else steps8(std::false_type{});
        // ---- boundary: next chunk's symbols land, then the staged words go out (lane `par` stores words 8 par .. 8 par + 7)
        land();
        asm volatile("" : "+v"(sy0), "+v"(sy1) : : "memory");
        if (cb > 8 * par) {
            const uint4 *src = reinterpret_cast<const uint4 *>(wb + 8 * par);
            typedef uint32_t u32x4_a4w __attribute__((ext_vector_type(4), aligned(4)));
            u32x4_a4w *dst = reinterpret_cast<u32x4_a4w *>(w + cnt + 8 * par);
            const uint4 a = src[0];
            dst[0] = u32x4_a4w{a.x, a.y, a.z, a.w};
            if (cb > 8 * par + 4) { const uint4 b2 = src[1]; dst[1] = u32x4_a4w{b2.x, b2.y, b2.z, b2.w}; }
        }
        cnt += cb;
        request((int32_t)C - 2);
    }
    if (live) { p->st[par] = s; if (par == 0) p->cnt = cnt; }
}

__global__ __launch_bounds__(64) void k_rans1_finish(const TileDesc *__restrict__ tiles, TileSel sel,
                                                     uint8_t *__restrict__ scratch2, const uint64_t *__restrict__ sbase2,
                                                     const uint32_t *__restrict__ stream_n, M2Blk *__restrict__ blk,
                                                     const W1Prep *__restrict__ prep, const uint16_t *__restrict__ wF) {
    const uint32_t tile = vtile(sel, blockIdx.x / M2_SLOTS), slot = blockIdx.x % M2_SLOTS, lane = threadIdx.x & 63;
    const W1Prep p = prep[(uint64_t)tile * M2_SLOTS + slot];
    if (p.kind != 1) return;
    const TileDesc t = tiles[tile];
    uint8_t *sc = scratch2 + sbase2[tile];
    const uint32_t n = stream_n[(uint64_t)tile * M2_SLOTS + slot];
    const uint32_t Nnom = m2_nominal(slot), N = p.N, distinct = p.distinct, cnt = p.cnt;
    const int pb = slot >= 17 ? 15 : 14;
    uint32_t *out = reinterpret_cast<uint32_t *>(sc + m2_off_blk(t.n, stream_n + (uint64_t)tile * M2_SLOTS, slot));
    M2Blk *mb = blk + (uint64_t)tile * M2_SLOTS + slot;
    if (lane < 2) { out[2 * lane] = (uint32_t)p.st[lane]; out[2 * lane + 1] = (uint32_t)(p.st[lane] >> 32); }  // state0 then state1 (libxpng.c:245)
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
        const uint16_t *F16 = wF + ((uint64_t)tile * M2_SLOTS + slot) * 256;
        BitW tb{0, 0, reinterpret_cast<uint32_t *>(sc + m2_off_piece(t.n, slot))};
        for (uint32_t k = 0; k < Nnom; k++) {
            const uint32_t F = k < N ? F16[k] : 0;
            if (!sparse) tb.put((uint32_t)pb, F);
            else if (F) tb.put((uint32_t)pb + 1, F + (1u << pb));
            else tb.put(1, 0);
        }
        tb.finish();
        *tb.p = 0;
        *mb = M2Blk{3u + (sparse ? 1u : 0u), n, cnt, tabBits};
    }
}

}  // namespace xpng
