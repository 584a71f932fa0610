// rans2_wide.hpp -- throughput form of the mode-1 entropy ENCODE stage: every lane runs a chain.
//
// k_rans2_encode (tile_container.hpp) gives one wavefront to one (tile, stream) pair, i.e. two live lanes out of 64.  That
// is the right shape for a single image (the stage is latency-bound by the longest chain), but a batch needs wave slots
// in proportion to tiles x streams.  Here the same block bytes are produced by three launches:
//
//   k_rans2_prep    wave per (tile, stream): histogram -> alphabet -> tables (rans_* helpers of rans2.hpp); empty and
//                   one-symbol blocks are finished on the spot; otherwise the encoder table and the normalised
//                   frequencies go to global memory
//   k_rans2_chain2  lane = one rANS state; a wave carries the same stream class of 16 (alpha) or 32 (context) tiles, so the
//                   chains of one wave have similar lengths; block-synchronous (see the kernel)
//   k_rans2_finish  wave per (tile, stream): states, header, frequency table, type-2 (raw) fallback
#pragma once
#include "common.hpp"
#include "rans2.hpp"

#include <type_traits>
namespace xpng {

constexpr uint32_t WTAB_TILE_BYTES = 4096 + 9 * 256;  // alpha table (256 x 16 B) + nine context tables (16 x 16 B)
struct WPrep {
    uint32_t kind;  // 0 = block already final (size in `cnt`), 1 = needs chain + finish
    uint32_t N, distinct, cnt;  // cnt: words emitted by the chain (kind 1)
    uint64_t st[2];             // final states
};
__device__ __forceinline__ uint32_t wtab_off(uint32_t c) { return c < 9 ? 4096 + c * 256 : 0; }
// [r4] The ALPHA encoder tables of a launch live interleaved, 32 streams (one chain wavefront's) to a group: entry e of the group's
// stream k at group * WATAB_GROUP_BYTES + e * 512 + k * 16.  A chain wavefront gathers one entry per lane and step; with a table per
// stream (rounds 3-4: 4 KB each, 6.4 KB apart) that is 64 different L1 lines per instruction whatever the symbols are - 767 M of a
// step's 4 249 M L1 accesses (profiles/r04_pmc_step_l2.json), and the compute unit's address path, one line per cycle, is what the chain
// wavefronts of 8 pipeline slots queue for at their block boundaries (tools/wave_probe.py).  Interleaved, the lanes
// whose symbol is the same - alpha residuals are mostly ONE value - read the same four lines.
constexpr uint32_t WATAB_GROUP_BYTES = 256u * 32u * 16u;
__host__ __device__ inline uint64_t watab_bytes(uint64_t streams) { return ((streams + 31) / 32 + 1) * WATAB_GROUP_BYTES; }
// (The alpha tables, 4 KB each, never enter LDS: the chain gathers the entries of a block from the table in global memory
//  - L2 - one block ahead.  Rounds 1-3 kept a 1 KB compact form per resident stream in LDS; LDS bytes x residency time summed
//  over the kernels of a pipelined step is what the step time tracks, and 162 waves x 37 KB x 26 ms was the
//  second largest item of that sum.)

// streams [c_first, c_first + c_count) of every tile (the alpha streams, c = 9, are prepared and chained while the stream-
// formation kernel is still producing the context streams: they only need the transform's alpha plane)
// (j0: the launch covers work items j0 .. of the enumeration - the alpha streams of the biggest tiles may be coded elsewhere, see
//  k_rans2_encode, tile_container.hpp)
__global__ __launch_bounds__(64) void k_rans2_prep(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t c_first, uint32_t c_count,
                                                   const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                   uint8_t *__restrict__ scratch, const uint32_t *__restrict__ ctx_n,
                                                   uint32_t *__restrict__ blk_sz, WPrep *__restrict__ prep,
                                                   uint8_t *__restrict__ wtab, uint16_t *__restrict__ wF, uint32_t j0,
                                                   uint8_t *__restrict__ watab) {
    bw_prio();
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cum[260];
    __shared__ EncSym tab[256];
    const uint32_t tile = vtile(sel, j0 + blockIdx.x / c_count), c = c_first + blockIdx.x % c_count, lane = threadIdx.x & 63;
    const TileDesc t = tiles[tile];
    uint8_t *sc = scratch + t.sbase;
    const uint8_t *in;
    uint32_t n, nominalN;
    int pb;
    if (c < 9) { in = sc + off_ctx(t.n, ctx_n + (uint64_t)tile * 9, (int)c); n = ctx_n[(uint64_t)tile * 9 + c]; nominalN = 9; pb = 12; }
    else { in = planes + 4 * plane_stride + t.pbase + 1; n = t.n - 1; nominalN = 256; pb = 15; }
    uint32_t *out = reinterpret_cast<uint32_t *>(sc + off_blk(t.n, ctx_n + (uint64_t)tile * 9, (int)c));
    WPrep *p = prep + (uint64_t)tile * 10 + c;
    if (n == 0) {  // libxpng.c:313
        if (lane == 0) { out[0] = 4; blk_sz[(uint64_t)tile * 10 + c] = 4; *p = WPrep{0, 0, 0, 4, {0, 0}}; }
        return;
    }
    rans_histogram(in, n, hist);
    uint32_t top, distinct;
    rans_alphabet(hist, nominalN, top, distinct);
    if (distinct == 1) {  // libxpng.c:318
        if (lane == 0) {
            out[0] = 8u | (1u << 24); out[1] = n | ((uint32_t)in[0] << 24);
            blk_sz[(uint64_t)tile * 10 + c] = 8; *p = WPrep{0, 1, 1, 8, {0, 0}};
        }
        return;
    }
    const uint32_t N = top + 1;
    rans_tables(hist, cum, tab, N, n, pb);
    EncSym *gt = reinterpret_cast<EncSym *>(wtab + (uint64_t)tile * WTAB_TILE_BYTES + wtab_off(c));
    uint16_t *gF = wF + ((uint64_t)tile * 10 + c) * 256;
    if (c == 9) {  // (alpha: the launch's interleaved layout; the stream's place in it is its work-item index, as the chain kernel counts it)
        const uint32_t jr = blockIdx.x / c_count;
        gt = reinterpret_cast<EncSym *>(watab + (uint64_t)(jr >> 5) * WATAB_GROUP_BYTES + (jr & 31u) * 16u);
        for (uint32_t i = lane; i < N; i += 64) { gt[i * 32u] = tab[i]; gF[i] = (uint16_t)hist[i]; }
    } else {
        for (uint32_t i = lane; i < N; i += 64) { gt[i] = tab[i]; gF[i] = (uint16_t)hist[i]; }
    }
    if (lane == 0) *p = WPrep{1, N, distinct, 0, {0, 0}};
}

// Block-synchronous form of the chain, packed by stream class (BIG: the alpha streams of 16 tiles, 32 lanes; else
// context stream c of 32 tiles, 64 lanes), so that the chains of one wave have similar lengths and a wave holds only the
// tables of its own class in LDS (8 KB for a context class).  Inside a block of 8 steps nothing touches global memory:
//   * the 8 symbols a lane codes in a block arrive as one aligned 16-byte load (+1 dword for the alpha plane, whose
//     symbols start at byte 1) issued one block earlier, and are picked out of registers (v_perm / v_bfe);
//   * renormalisation words are staged in LDS in emission order (state0's word first) and leave at the block boundary as
//     16-byte stores.  Words past the pair's count are scratch: the next block, or k_rans2_finish, overwrites them.
// So no s_waitcnt for a global access sits in the dependent chain (a conditional store per step does exactly that on
// gfx9, where stores count on vmcnt).
typedef uint32_t u32x4_enc __attribute__((ext_vector_type(4)));
template <bool BIG> constexpr uint32_t chain2_ltab_bytes() { return BIG ? 0u : 32u * (144u + 16u); }
// (Round 3 tried the context class with its symbols requested THREE blocks ahead through an LDS ring filled by LDS-DMA loads -
//  global_load_lds_dwordx4, the mechanism of tools/ubench/ldsdma.hip - bit-exact, 60 GPU tests green: 6.5 -> 7.5 ms alone, 11.2 ms
//  median in the pipeline before and after.  This chain does not wait for its symbols there; removed.  DESIGN.md 6.2b.)
template <bool BIG> constexpr size_t chain2_lds_bytes() { return chain2_ltab_bytes<BIG>() + 32u * WB_STRIDE * 4u; }  // dynamic LDS of one launch (common.hpp: why dynamic)
template <bool BIG>
__global__ __launch_bounds__(64) void k_rans2_chain2(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t total,
                                                     const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                     uint8_t *__restrict__ scratch, const uint32_t *__restrict__ ctx_n,
                                                     WPrep *__restrict__ prep, const uint8_t *__restrict__ wtab, uint32_t j0,
                                                     const uint8_t *__restrict__ watab) {
    constexpr uint32_t TPW = 32;                  // tiles (streams) per wave: every lane carries a state (alpha ran 16 per wave while its tables took 4 KB of LDS each)
    // bytes of one encoder table in LDS: a context stream has the nine symbols nl = 0..8 (the alpha class keeps none there)
    constexpr uint32_t TAB = 144;
    constexpr uint32_t TSTRIDE = TAB + 16;        // +4 banks per table: lanes mostly look up the same symbol
    constexpr uint32_t SH = BIG ? 1 : 0;          // byte phase of the symbols inside 16-byte chunks (alpha symbol of pixel i is plane byte i)
    constexpr int PB = BIG ? 15 : 12;
    static_assert((BIG || TPW * TSTRIDE == chain2_ltab_bytes<BIG>()) && (TPW * TSTRIDE) % 16 == 0, "LDS layout");
    extern __shared__ __align__(16) uint8_t chain2_lds[];
    uint8_t *const ltab = chain2_lds;                                                               // [TPW * TSTRIDE] (context class only)
    uint32_t *const wbuf = reinterpret_cast<uint32_t *>(chain2_lds + chain2_ltab_bytes<BIG>());     // [TPW * WB_STRIDE] per stream: 16 staged words + dump words nobody reads (common.hpp)
    __builtin_amdgcn_s_setprio(XPNG_CHAIN_PRIO);
    XPNG_PROBE_BEGIN()
    const uint32_t lane = threadIdx.x & 63, k = lane >> 1, par = lane & 1;
    const uint32_t c = BIG ? 9 : blockIdx.x % 9, grp = BIG ? blockIdx.x : blockIdx.x / 9;
    const uint32_t j = j0 + grp * TPW + k;
    bool live = k < TPW && j < total;
    if (!BIG) {
        for (uint32_t ts = 0; ts < TPW; ts++) {
            const uint32_t jj = j0 + grp * TPW + ts;
            if (jj >= total) break;
            const uint4 *src = reinterpret_cast<const uint4 *>(wtab + (uint64_t)vtile(sel, jj) * WTAB_TILE_BYTES + wtab_off(c));
            uint4 *dst = reinterpret_cast<uint4 *>(ltab + ts * TSTRIDE);
            for (uint32_t i = lane; i < TAB / 16; i += 64) dst[i] = src[i];
        }
        __syncthreads();
    }
    const uint32_t tile = vtile(sel, live ? j : 0);
    const TileDesc *t = tiles + tile;
    uint8_t *sc = scratch + t->sbase;
    WPrep *p = prep + (uint64_t)tile * 10 + c;
    live = live && p->kind == 1;
    const uint8_t *in = BIG ? planes + 4 * plane_stride + t->pbase : sc + off_ctx(t->n, ctx_n + (uint64_t)tile * 9, BIG ? 0 : (int)c);  // 16-byte aligned; symbol i at in[i + SH]
    const uint32_t n = !live ? 0 : BIG ? t->n - 1 : ctx_n[(uint64_t)tile * 9 + c];
    const uint32_t mysteps = (n + 1 - par) >> 1;  // state0 codes ceil(n/2) symbols, state1 floor(n/2)
    uint32_t T = mysteps, Tmin = live ? mysteps : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t v = __shfl_xor(T, o), v2 = __shfl_xor(Tmin, o);
        T = v > T ? v : T; Tmin = v2 < Tmin ? v2 : Tmin;
    }
    T = sgpr((T + 7) & ~7u); Tmin = sgpr(Tmin);
    const EncSym *tab = reinterpret_cast<const EncSym *>(ltab + (k < TPW ? k : 0) * TSTRIDE);
    uint32_t *w = reinterpret_cast<uint32_t *>(sc + off_blk(t->n, ctx_n + (uint64_t)tile * 9, (int)c)) + 3;
    uint32_t *wb = wbuf + (k < TPW ? k : 0) * WB_STRIDE;
    const uint32_t dumpw = wb_dump(k < TPW ? k : 0, par);
    // symbols of block b (pair symbols 16b .. 16b+15) = bytes [16b + SH, 16b + SH + 16) of `in`
    u32x4_enc fq = u32x4_enc{0, 0, 0, 0};  // in flight: the block after the current one (ONE 128-bit value: four 32-bit loop-carried values
    uint32_t f4 = 0;                       // get registers of their own and are copied out of the load's tuple behind a wait for it)
    // (unconditional, on a clamped block index: a branch around the loads makes the compiler keep the loaded values in
    //  temporaries and copy them into the loop-carried registers right away, and that copy waits for the load it follows)
    const uint32_t last_blk = mysteps ? (mysteps - 1) >> 3 : 0;
    auto request = [&](uint32_t b) __attribute__((always_inline)) {
        const uint8_t *a = in + 16ull * (b < last_blk ? b : last_blk);
        fq = *reinterpret_cast<const u32x4_enc *>(a);
        if (SH) f4 = *reinterpret_cast<const uint32_t *>(a + 16);
    };
    uint32_t sy0 = 0, sy1 = 0;  // this lane's 8 symbols of the current block, one per byte
    auto land = [&]() __attribute__((always_inline)) {
        const uint32_t f0 = fq.x, f1 = fq.y, f2 = fq.z, f3 = fq.w;
        uint32_t d0 = f0, d1 = f1, d2 = f2, d3 = f3;
        if (SH) {
            d0 = __builtin_amdgcn_alignbyte(f1, f0, SH); d1 = __builtin_amdgcn_alignbyte(f2, f1, SH);
            d2 = __builtin_amdgcn_alignbyte(f3, f2, SH); d3 = __builtin_amdgcn_alignbyte(f4, f3, SH);
        }
        const uint32_t selb = par ? 0x07050301u : 0x06040200u;  // odd / even bytes of a dword pair
        sy0 = __builtin_amdgcn_perm(d1, d0, selb);
        sy1 = __builtin_amdgcn_perm(d3, d2, selb);
    };
    // BIG: the 8 table entries of a block are fetched ONE BLOCK AHEAD (its symbols land two blocks ahead) from the stream's
    // table in global memory (global address space: a pointer computed from a kernel argument through a struct would be
    // generic, and flat loads also count on lgkmcnt); no table access inside a block, and every load has a whole block to return
    typedef const __attribute__((address_space(1))) u32x4_enc *gent;
    // (the stream's column of its group's interleaved table: entry e at gfull[e * 32]; lanes without a stream read their own, unwritten
    //  column - inside the allocation, and their steps are inactive)
    const gent gfull = (gent)(uintptr_t)(watab + (uint64_t)grp * WATAB_GROUP_BYTES + (k & 31u) * 16u);
    constexpr uint32_t cmpl_base = 1u << PB;
    constexpr int thr_shift = 31 - PB;
    uint64_t s = RANS_L;
    uint32_t cnt = 0;
    // BIG: two sets of eight entries, X and Y, used ALTERNATELY: a block steps on one set and refills that same set at its boundary (for
    // the block after next, whose symbols have just landed) while the next block steps on the other.  [r4] Rounds 3-4 shipped "E = the
    // current block's entries, En = the next block's, E = En at the boundary": the compiler treats that copy as a loop-carried value and
    // sinks it to the loop's back edge, behind the gathers - En's old and new values are then alive together, the gathers land in
    // temporaries, and the back edge copies those into En behind an s_waitcnt vmcnt(0): every block waited for the loads it had just
    // issued (a wavefront spent 12 % of its life in this boundary alone, 26 % at 8 pipeline slots: tools/wave_probe.py), and 32 register
    // moves per block came on top.  With two sets and the loop unrolled twice nothing is copied and a gather has a whole block to land.
    EncSym X[8], Y[8];
    auto entries = [&](EncSym (&A)[8]) __attribute__((always_inline)) {  // entries of the block whose symbols are in sy0/sy1 -> A
        uint32_t sy[8];
#pragma unroll
        for (int u = 0; u < 8; u++) sy[u] = ((u < 4 ? sy0 : sy1) >> (8 * (u & 3))) & 255u;
        // (lanes without a stream read their placeholder tile's table: no branch around the loads)
        // [r4] tried: the entries of the stream's two most frequent symbols in registers and only the other lanes gathering (behind
        // s_cbranch_execz; bit-exact, 95 % of the gathers gone).  A conditional load makes the number of loads in flight unknown to the
        // compiler, so every wait behind one becomes vmcnt(0): the block that follows waits for the gathers just issued.  (In the
        // E / En form of rounds 3-4: 22.3 instead of 19.3 ms alone; as inline assembly on the destination registers the compiler copies
        // those registers, unlanded, at the back edge.)  profiles/r04_experiments.txt.
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32x4_enc v = gfull[sy[u] * 32u]; A[u] = EncSym{v.x, v.y, v.z, v.w}; }
    };
    request(0); land(); request(1);
    EncSym e = EncSym{0, 0, 0, 0};
    if (BIG) {
        entries(X);                       // X = entries of block 0
        land(); request(2); entries(Y);   // symbols of block 1 land, block 2 is requested, Y = entries of block 1 (in flight)
    } else e = tab[sy0 & 255u];
    // one block: eight steps on the entries A (BIG), then the boundary
    auto block = [&](const uint32_t kb, EncSym (&A)[8]) __attribute__((always_inline)) {
        uint32_t cb = 0;  // words the pair has staged in this block
        // eight steps; FULL (compile time): every lane of the wave is inside its stream for the whole block, no per-lane activity test
        auto steps8 = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                EncSym en = e;
                if (BIG) e = A[u];
                else if (u < 7) en = tab[((u + 1 < 4 ? sy0 : sy1) >> (8 * ((u + 1) & 3))) & 255u];  // entry of the NEXT step, under this step's arithmetic
                const bool act = FULL || kb + (uint32_t)u < mysteps;
                const uint32_t freq = e.freq_shift & 0xFFFF, rsh = e.freq_shift >> 16;
                const uint32_t emit = (act && (uint32_t)(s >> 32) >= (freq << thr_shift)) ? 1u : 0u;
                // the spill handling (pair swap, LDS store, state shift, count) sits behind a wave-uniform branch: the alpha streams
                // of a wave spill in about a third of the steps (a state spills once in ~160), so most steps are the arithmetic only
                if (!BIG || __ballot(emit)) {
                    const uint32_t other = swap_pair(emit);
                    wb[emit ? cb + (par ? other : 0u) : dumpw] = (uint32_t)s;  // state0's word first (libxpng.c:370-373)
                    if (emit) s >>= 32;
                    cb += emit + other;
                }
                if (act) {
                    const uint64_t rcp = ((uint64_t)e.rcp_hi << 32) | e.rcp_lo;
                    const uint64_t q = __umul64hi(s, rcp) >> rsh;
                    if (BIG) {
                        // (the bias is added with a carry chain: as a 64-bit add the compiler wants it in a register PAIR (bias, 0), copies
                        //  it there out of the gathered entry at the loop's back edge - and waits for the gathers it has just issued)
                        // (q = s / freq < 2^48 behind the spill test and cmpl <= 2^15: the high half of q x cmpl is one 24-bit multiply-add
                        //  into the high word; as a 64 x 32 multiply the compiler builds a (word, 0) register pair for a second 64-bit one)
                        const uint32_t cm = cmpl_base - freq;
                        const uint64_t t = s + (uint64_t)(uint32_t)q * cm;
                        uint32_t tl = (uint32_t)t, th = (uint32_t)(t >> 32) + __umul24((uint32_t)(q >> 32), cm);
                        asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(tl), "+v"(th) : "v"(e.bias) : "vcc");
                        s = ((uint64_t)th << 32) | tl;
                    } else s += e.bias + q * (uint64_t)(cmpl_base - freq);
                }
                if (!BIG) e = en;
            }
        };
        if (kb + 8 <= Tmin) steps8(std::true_type{});
        else steps8(std::false_type{});
        // ---- boundary: the next block's symbols land FIRST (the wait in front of it then covers the load issued a block ago
        // and nothing younger: behind this block's stores it would also wait for their acknowledgement), then the staged
        // words go out (lane `par` stores words 8 par .. 8 par + 7) and the block after next is requested
        XPNG_PROBE_WAIT()
        XPNG_PROBE_ISSUE_BEGIN()
        land();
        asm volatile("" : "+v"(sy0), "+v"(sy1) : : "memory");  // (pins the landing in front of the stores)
        if (cb > 8 * par) {
            const uint4 *src = reinterpret_cast<const uint4 *>(wb + 8 * par);
            typedef uint32_t u32x4_a4w __attribute__((ext_vector_type(4), aligned(4)));
            u32x4_a4w *dst = reinterpret_cast<u32x4_a4w *>(w + cnt + 8 * par);
            const uint4 a = src[0];
            dst[0] = u32x4_a4w{a.x, a.y, a.z, a.w};
            if (cb > 8 * par + 4) { const uint4 b2 = src[1]; dst[1] = u32x4_a4w{b2.x, b2.y, b2.z, b2.w}; }
        }
        cnt += cb;
        request((kb >> 3) + (BIG ? 3 : 2));
        if (BIG) entries(A);   // the set this block has just finished with: entries of the block after next
        else e = tab[sy0 & 255u];
        XPNG_PROBE_ISSUE_END()
    };
    if (BIG) {
        for (uint32_t kb = 0; kb < T; kb += 16) {
            block(kb, X);
            if (kb + 8 >= T) break;
            block(kb + 8, Y);
        }
    } else {
        for (uint32_t kb = 0; kb < T; kb += 8) block(kb, X);
    }
    if (live) { p->st[par] = s; if (par == 0) p->cnt = cnt; }
    XPNG_PROBE_END(BIG ? 2 : 1)
}

__global__ __launch_bounds__(64) void k_rans2_finish(const TileDesc *__restrict__ tiles, TileSel sel, uint32_t spt,
                                                     const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                     uint8_t *__restrict__ scratch, const uint32_t *__restrict__ ctx_n,
                                                     uint32_t *__restrict__ blk_sz, const WPrep *__restrict__ prep,
                                                     const uint16_t *__restrict__ wF) {
    bw_prio();
    const uint32_t tile = vtile(sel, blockIdx.x / spt), c = blockIdx.x % spt, lane = threadIdx.x & 63;
    const WPrep p = prep[(uint64_t)tile * 10 + c];
    if (p.kind != 1) return;
    const TileDesc t = tiles[tile];
    uint8_t *sc = scratch + t.sbase;
    const uint8_t *in;
    uint32_t n;
    int pb;
    if (c < 9) { in = sc + off_ctx(t.n, ctx_n + (uint64_t)tile * 9, (int)c); n = ctx_n[(uint64_t)tile * 9 + c]; pb = 12; }
    else { in = planes + 4 * plane_stride + t.pbase + 1; n = t.n - 1; pb = 15; }
    uint8_t *out8 = sc + off_blk(t.n, ctx_n + (uint64_t)tile * 9, (int)c);
    uint32_t *out = reinterpret_cast<uint32_t *>(out8);
    const uint16_t *F16 = wF + ((uint64_t)tile * 10 + c) * 256;
    const uint32_t N = p.N, distinct = p.distinct, rawBits = (uint32_t)bit_width(N - 1);
    uint32_t *w = out + 3 + p.cnt;
    if (lane < 2) { w[2 * lane] = (uint32_t)p.st[lane]; w[2 * lane + 1] = (uint32_t)(p.st[lane] >> 32); }  // libxpng.c:394
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
            const uint32_t F = F16[k];
            if (!sparse) tb.put((uint32_t)pb, F);
            else if (F) tb.put((uint32_t)pb + 1, F + (1u << pb));
            else tb.put(1, 0);
        }
        tb.finish();
        csz = (uint32_t)((uint8_t *)tb.p - out8);
        out[0] = csz | ((3u + (sparse ? 1u : 0u)) << 24);
    }
    csz = __shfl(csz, 0);
    // ---- raw fallback, type 2 (libxpng.c:417-424)
    const uint64_t rawTotalBits = (uint64_t)rawBits * n;
    const uint32_t rawWords = (uint32_t)((rawTotalBits + 31) >> 5);
    if (csz >= 8 + 4 * rawWords) {
        __syncthreads();
        for (uint32_t wi = lane; wi < rawWords; wi += 64) {
            const uint64_t b0 = (uint64_t)wi * 32;
            uint32_t jx = (uint32_t)(b0 / rawBits);
            uint32_t word = 0;
            for (; jx < n; jx++) {
                const int64_t rel = (int64_t)((uint64_t)jx * rawBits) - (int64_t)b0;
                if (rel >= 32) break;
                const int sh = 32 - (int)rel - (int)rawBits;
                const uint32_t v = in[jx];
                word |= sh >= 0 ? (sh < 32 ? v << sh : 0u) : v >> (-sh);
            }
            out[2 + wi] = word;
        }
        csz = 8 + 4 * rawWords;
        if (lane == 0) { out[0] = csz | (2u << 24); out[1] = n | (rawBits << 24); }
    }
    if (lane == 0) blk_sz[(uint64_t)tile * 10 + c] = csz;
}

}  // namespace xpng
