// common.hpp -- shared device/host definitions of the MI355X xPNG tile codec (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace xpng {

// ---- build flavours ---------------------------------------------------------------------------------------------------
// The release library (libxpng_hip.so) reads exactly these five environment variables, each of which selects between forms that
// produce the SAME bytes (INTEGRATION.md lists them): XPNG_DEVICE, XPNG_GPUS, XPNG_WIDE_RANS, XPNG_NARROW_RANS,
// XPNG_NO_SPLIT (it also WRITES the runtime's GPU_MAX_HW_QUEUES when it is loaded, unless the caller has set it).  Everything that exists for timing
// studies - kernel knock-outs, unused-LDS pads, no-store switches, phase stamps, the wave probe, stream priorities, fake
// devices - is compiled only into libxpng_hip_probes.so (-DXPNG_PROBES, `make probes`; tools/ load that one): a product
// library whose output can be falsified through the environment is not shippable.
#ifdef XPNG_PROBES
inline const char *probe_env(const char *name) { return getenv(name); }
#else
inline const char *probe_env(const char *) { return nullptr; }
#endif
inline size_t probe_pad(const char *name) { const char *v = probe_env(name); return v ? (size_t)atoi(v) : 0; }  // bytes of unused dynamic LDS (occupancy throttle)

// Streams that carry serial-chain kernels.  tools/wave_probe.py shows that in the pipelined bench the chain WAVES run at their solo
// speed while the chain KERNELS take 1.7x longer: their workgroups (tens of KB of LDS each) trickle onto CUs that bandwidth kernels
// with 10^4..10^5 queued workgroups keep full.  Stream priority is the obvious lever and the wrong one (measured: 27.8 against
// 39.2 Gpx/s with the highest priority - the normal-priority queues starve while a priority kernel runs), so the priority is 0
// outside probe builds.
inline hipError_t chain_stream_create(hipStream_t *s) {
    int lo = 0, hi = 0;
    // (probe builds) XPNG_SIDE_CUMASK=K: the side streams of a context - alpha encode chains, alpha decode chains, the small-tile decode
    // tail - may use only the first K compute units of the runtime's numbering (the driver deals a queue's mask bits round-robin over
    // the shader engines and XCDs: K / 8 per XCD); the caller's stream keeps all of them.  VERDICT r3 item 1a: a CU partition.
    if (const char *m = probe_env("XPNG_SIDE_CUMASK")) {
        const int k = atoi(m);
        if (k > 0 && k < 256) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < k; i++) mask[i >> 5] |= 1u << (i & 31);
            return hipExtStreamCreateWithCUMask(s, 8, mask);
        }
    }
    if (probe_env("XPNG_STREAM_PRIORITY")) (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, probe_env("XPNG_STREAM_PRIORITY") ? hi : 0);
}

// Wave probe for placement studies (tools/wave_probe.py; probe builds only): when a buffer is registered (xpnghip_debug_probe),
// wave 0 of every workgroup of the serial-chain kernels records where it ran (HW_ID: SE / CU / SIMD / wave slot; XCC_ID) and
// when (constant 100 MHz clock); bit 4 of `xcc` clear: its bits 5.. carry the wave's shader-clock cycles / 16 (s_memtime), i.e. the clock it ran at; bit 4 set: its upper 16 bits the fraction (x 65536) of its life between XPNG_PROBE_ISSUE_BEGIN and _END; the upper 16 bits of `block` the fraction (x 65536) of its life it stood in XPNG_PROBE_WAIT.
struct WaveProbe { uint32_t kernel, block, hwid, xcc; uint64_t t0, t1; };
#ifdef XPNG_PROBES
__device__ WaveProbe *g_probe_buf = nullptr;
__device__ uint32_t g_probe_cap = 0, g_probe_n = 0;
#define XPNG_PROBE_BEGIN()                                                                        \
    WaveProbe *const probe_buf_ = g_probe_buf;                                                    \
    uint64_t probe_t0_ = 0, probe_c0_ = 0, probe_wait_ = 0, probe_iss_ = 0;                       \
    [[maybe_unused]] uint64_t probe_ia_ = 0;                                                      \
    if (probe_buf_) { probe_t0_ = __builtin_amdgcn_s_memrealtime(); probe_c0_ = __builtin_readcyclecounter(); }
// at a block boundary of a chain kernel, where the code is about to use what the previous boundary requested (the youngest loads in
// flight, so the wait it runs into is vmcnt(0) anyway): how long does the wavefront stand there?  (100 MHz ticks, summed)
#define XPNG_PROBE_WAIT()                                                                         \
    if (probe_buf_) {                                                                             \
        const uint64_t pwa_ = __builtin_amdgcn_s_memrealtime();                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        probe_wait_ += __builtin_amdgcn_s_memrealtime() - pwa_;                                   \
    }
// ... and how long does the rest of the boundary take - landing the words, the stores, ISSUING the next requests (a vector memory
// instruction that finds the compute unit's address path full stalls the wavefront at issue, whatever its prefetch distance)?
#define XPNG_PROBE_ISSUE_BEGIN() if (probe_buf_) { asm volatile("" ::: "memory"); probe_ia_ = __builtin_amdgcn_s_memrealtime(); }
#define XPNG_PROBE_ISSUE_END() if (probe_buf_) { asm volatile("" ::: "memory"); probe_iss_ += __builtin_amdgcn_s_memrealtime() - probe_ia_; }
#define XPNG_PROBE_END(kid)                                                                       \
    if (probe_buf_ && threadIdx.x == 0) {                                                         \
        const uint32_t pi_ = atomicAdd(&g_probe_n, 1u);                                           \
        if (pi_ < g_probe_cap)                                                                    \
            probe_buf_[pi_] = WaveProbe{(uint32_t)(kid),                                                                           \
                                        (blockIdx.x & 0xFFFFu) | ((uint32_t)((probe_wait_ << 16) / ((__builtin_amdgcn_s_memrealtime() - probe_t0_) | 1)) << 16), \
                                        (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4),                                       \
                                        ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) |                          \
                                            (probe_iss_ ? ((uint32_t)((probe_iss_ << 16) / ((__builtin_amdgcn_s_memrealtime() - probe_t0_) | 1)) << 16) | 0x10u \
                                                        : ((uint32_t)((__builtin_readcyclecounter() - probe_c0_) >> 4) << 5)),   \
                                        probe_t0_, __builtin_amdgcn_s_memrealtime()};                                             \
    }
#else
#define XPNG_PROBE_BEGIN()
#define XPNG_PROBE_WAIT()
#define XPNG_PROBE_ISSUE_BEGIN()
#define XPNG_PROBE_ISSUE_END()
#define XPNG_PROBE_END(kid)
#endif

// VALU burner (probe builds only): XPNG_BURN_TR=k / XPNG_BURN_ST=k make every thread of the transform / routing kernels issue 4 k
// extra full-rate vector instructions per workgroup / per iteration.  Is the pipelined step bound by vector instruction issue?  If
// it is, `value` falls in proportion to the instructions added (profiles/r04_burn.txt).
#ifdef XPNG_PROBES
__device__ uint32_t g_burn_tr = 0, g_burn_st = 0;
__device__ __forceinline__ void probe_burn(uint32_t k, uint32_t seed, uint32_t *sink) {
    if (!k) return;
    uint32_t x = seed;
    for (uint32_t i = 0; i < k; i++) {
        x = (x ^ (x << 3)) + i;   // v_lshlrev + v_xor (or one v_lshl_xor?) + v_add: dependent, full rate
        x = (x ^ (x >> 5)) + seed;
    }
    if (x == 0x5A17C0DEu) *sink = x;  // (never, in practice: keeps the loop alive)
}
#define XPNG_BURN(var, seed, sink) probe_burn(var, seed, sink)
#else
#define XPNG_BURN(var, seed, sink)
#endif

// Knock-out switch for timing studies (tools/knockout.py; probe builds only): XPNG_SKIP=name,name,... leaves the named kernels
// of the batched level-1 paths unlaunched once XPNG_SKIP_AFTER launch sequences have run complete (the workspaces then still
// hold the previous, identical results, so everything downstream keeps working on valid data).
#ifdef XPNG_PROBES
inline std::atomic<uint64_t> &dbg_sequences() { static std::atomic<uint64_t> n{0}; return n; }  // (callers may come from several host threads)
inline void dbg_count_sequence() { dbg_sequences().fetch_add(1, std::memory_order_relaxed); }
inline bool dbg_skip(const char *name) {
    static const char *list = getenv("XPNG_SKIP");
    if (!list) return false;
    static const uint64_t after = getenv("XPNG_SKIP_AFTER") ? strtoull(getenv("XPNG_SKIP_AFTER"), nullptr, 10) : 16;
    if (dbg_sequences().load(std::memory_order_relaxed) <= after) return false;
    const size_t n = strlen(name);
    for (const char *p = list; (p = strstr(p, name)); p += n)
        if ((p == list || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
#else
inline void dbg_count_sequence() {}
constexpr bool dbg_skip(const char *) { return false; }
#endif

// LDS of the serial-chain kernels is DYNAMIC (extern __shared__, size passed at launch), never a static array.  With a static
// array the compiler knows that 37-51 KB per single-wave workgroup allows at most one such wave per SIMD and then RAISES the
// kernel's register allocation to the smallest number that enforces exactly that (AMDGPU: NumVGPRsForWavesPerEU = 257 for a
// kernel that uses 82 - "Occupancy: 1" in the ISA listing): every alpha-chain / walk wave then held 264 of its SIMD's 512
// registers, no two of them could share a SIMD, and a SIMD that hosted one had room for three transform waves instead of
// five.  That is the "fat workgroups start 5-15 ms late" of profiles/r02_wave_probe_p4.txt and most of why bandwidth kernels
// ran 5-8x their solo time beside the chains.  With dynamic LDS the allocation is what the code uses.
// wave priority of the serial-chain kernels (s_setprio; the throughput kernels run at 0)
#ifndef XPNG_CHAIN_PRIO
#define XPNG_CHAIN_PRIO 3
#endif
// wave priority of the chip-filling (throughput) kernels of the batched paths, set by their first instruction (r4 experiment:
// with everything at 0 the issue arbiter serves the OLDEST wave first, i.e. the long-lived chain waves, just as with the chains at 3)
#ifndef XPNG_BW_PRIO
#define XPNG_BW_PRIO 0
#endif
__device__ __forceinline__ void bw_prio() {
#if XPNG_BW_PRIO
    __builtin_amdgcn_s_setprio(XPNG_BW_PRIO);
#endif
}
constexpr uint32_t TILE_AREA = 444u * 444u;  // reference libxpng.c:49
constexpr uint32_t NL_NONE = 0xFFu;          // nl-plane marker: pixel emits no colour symbol
constexpr int WAVE = 64;

// One tile of the image (reference task_t, libxpng.c:46) plus where its intermediates live.
struct TileDesc {
    uint32_t x, y, w, h;  // top-left pixel and size inside the raster
    uint32_t n;           // w*h
    uint32_t img;         // image index inside the batch (rasters[img], blobs[img])
    uint64_t pbase;       // first index of this tile in each symbol plane (multiple of 256)
    uint64_t sbase;       // byte offset of this tile's stream scratch (multiple of 256)
};

// A launch covers tiles [t0, t0 + cnt) of EVERY image of the batch (nimg images).  Work item j in [0, nimg*cnt) maps to
// entry vtile(j) of the (B * N)-entry tile table:
//   order == nullptr : image-major, image j / cnt, tile t0 + j % cnt
//   order != nullptr : tile-major over order[] (the cnt tile indices sorted by decreasing pixel count): tile order[j / nimg],
//                      image j % nimg.  Neighbouring work items then have chains of equal length (the wide kernels put
//                      32-64 of them in one wavefront, which runs for its longest), and the biggest tiles come first.
// Per-tile outputs that the host sees per image (sizes, offsets) are indexed by imglin() = image * cnt + (tile - t0).
struct TileSel { uint32_t t0, cnt, N, nimg; const uint32_t *order; };
__host__ __device__ inline uint32_t vtile(const TileSel &s, uint32_t j) {
    if (s.order) { const uint32_t r = j / s.nimg; return (j - r * s.nimg) * s.N + s.order[r]; }
    return (j / s.cnt) * s.N + s.t0 + (j % s.cnt);
}
__host__ __device__ inline uint32_t imglin(const TileSel &s, uint32_t vt) { const uint32_t img = vt / s.N; return img * s.cnt + (vt - img * s.N - s.t0); }

// Stream-scratch layout of one tile, all offsets relative to TileDesc::sbase:
//   [k bit stream: 8*pxsz bits + <= 24 bits/pixel][alpha block][nine context streams, back to back][nine context blocks, back to back]
// The k region and the alpha block are sized for the worst case (their places must not depend on anything computed on the
// device: the alpha chains start while the routing kernel is still running).  The nine context streams share the tile's
// n - 1 coded pixels, and how they share them is known BEFORE the routing kernel writes a byte: stream c receives the nl of
// every coded pixel whose predecessor's nl is c, so its length is hist[c] - [c == nl of the last coded pixel] + [c == 0], a
// histogram of the nl plane (taken by the transform as it writes the plane: nlacc_* / k_m1_lens, m1_encode.hpp).  So each stream gets the room its length needs (+ slack for the 16-byte block loads
// of the chains) and the region is n + 9 * 96 bytes, not nine times n: 7.5 instead of 15.5 bytes of scratch per pixel.
//   rANS v2 block   : 12 B header + ceil(m*pb/32) words + 16 B states + table (<= 256*16 bits), m = symbols of the stream
__host__ __device__ inline uint64_t rup(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
__host__ __device__ inline uint64_t kw_cap(uint32_t n) { return rup(3ull * n + 64, 256); }
__host__ __device__ inline uint64_t ctxblk_cap(uint32_t n) { return rup(3ull * n / 2 + 128, 256); }
__host__ __device__ inline uint64_t alphablk_cap(uint32_t n) { return rup(2ull * n + 1280, 256); }
__host__ __device__ inline uint64_t ctx_slot(uint32_t m) { return rup((uint64_t)m + 32, 64); }       // room of a context stream of m symbols
__host__ __device__ inline uint64_t ctx_region(uint32_t n) { return rup((uint64_t)n + 9 * 96, 256); }  // >= sum of the nine slots
__host__ __device__ inline uint64_t off_kw(uint32_t) { return 0; }
// cn: the tile's nine context stream lengths (ctx_n + tile * 9)
__host__ __device__ inline uint64_t off_ctx(uint32_t n, const uint32_t *cn, int c) {
    uint64_t o = kw_cap(n) + alphablk_cap(n);
    for (int i = 0; i < c; i++) o += ctx_slot(cn[i]);
    return o;
}
// Block slots: the alpha block right behind k, the nine context blocks behind the stream region, each sized by ITS stream's
// length: the streams share n - 1 symbols, so the nine take <= 1.5 n + 9 * 383 bytes together, not 13.5 n.
__host__ __device__ inline uint64_t off_blk(uint32_t n, const uint32_t *cn, int c) {  // c = 0..8 context blocks, 9 = alpha
    uint64_t o = kw_cap(n);
    if (c == 9) return o;
    o += alphablk_cap(n) + ctx_region(n);
    for (int i = 0; i < c; i++) o += ctxblk_cap(cn[i]);
    return o;
}
__host__ __device__ inline uint64_t tile_scratch_bytes(uint32_t n) {
    return kw_cap(n) + alphablk_cap(n) + ctx_region(n) + rup(3ull * n / 2 + 9 * 383, 256);
}

// ---- integer helpers shared by encode and decode (reference libxpng.c:19-30) ----------------------
__device__ __forceinline__ int bit_width(uint32_t v) { return v ? 32 - __clz((int)v) : 0; }  // numBit
__device__ __forceinline__ int zz_enc(int d) {                                                 // pix_toU
    int v = (int)(int8_t)d;
    return ((v << 1) ^ (v >> 31)) & 0xFF;
}
__device__ __forceinline__ int zz_dec(int u) { return (u >> 1) ^ -(u & 1); }                   // pix_toS
__device__ __forceinline__ int pred_avg(int L, int U) { return (L + U + 1) >> 1; }            // p2a
__device__ __forceinline__ int pred_grad(int L, int U, int UL) { return ((3 * L + 3 * U - 2 * UL) + 2) >> 2; }  // p3a

// 3 x as ONE full-rate instruction (x + (x << 1)): the compiler turns a multiplication of a 25-bit value by 3 into v_mul_lo_u32, which
// issues at a quarter of the vector rate (the gradient predictor does it for every pixel, on both sides of the codec)
__device__ __forceinline__ uint32_t times3(uint32_t x) {
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    uint32_t lane = threadIdx.x & 63u;
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}

// predictor flags from the four sampled cost sums: first minimum wins (libxpng.c:133-139);
// tiles narrower than 4 px get 0 (libxpng.c:94)
__device__ __forceinline__ int pr_from_sums(const uint32_t *s, int pxsz, uint32_t tw, uint32_t th) {
    if (tw < 4 || th < 4) return 0;
    int m = 0;
    uint32_t r = s[0];
    if (s[1] < r) { m = 1; r = s[1]; }
    if (s[2] < r) { m = 2; r = s[2]; }
    if (s[3] < r) { m = 3; r = s[3]; }
    return (pxsz & 4) | m;
}

// packed pixel load: r | g<<8 | b<<16 (| a<<24).  Rasters are 4-byte aligned, so is every RGBA pixel.
template <int PXSZ>
__device__ __forceinline__ uint32_t load_px(const uint8_t *p) {
    if constexpr (PXSZ == 4) {
        return *reinterpret_cast<const uint32_t *>(p);
    } else {
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
}

constexpr uint64_t RANS_L = 1ull << 31;  // reference RANS64_L, libxpng.c:153

// Word staging of the wide ENCODE chains (k_rans2_chain2, k_rans1_chain): every stream of a wave owns WB_STRIDE words of LDS -
// 16 staged words and a dump word for the lanes that emit nothing in a step, which is most lanes in most steps.  The stride
// and the dump word's place are chosen by LDS BANK: with 32 words per stream (rounds 1-3) all 64 lanes of a step's store hit
// banks 16 and 17 - a 32-deep serialised store per step and wave, 16-26 conflict cycles per LDS instruction in the counters,
// and every LDS access of the compute unit queues behind it.  With 36 words per stream (a multiple of 16 bytes: the staged
// words leave as 16-byte reads) and the dump word at 16 + 2 (stream / 8) + parity the 64 lanes fall on 32 banks, two deep.
constexpr uint32_t WB_STRIDE = 36;
__device__ __forceinline__ uint32_t wb_dump(uint32_t stream, uint32_t par) { return 16u + 2u * (stream >> 3) + par; }

// swap values between lanes 2k and 2k+1 (DPP quad_perm [1,0,3,2]); VALU only, no LDS round trip
__device__ __forceinline__ uint32_t swap_pair(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// inclusive prefix sum over the 64 lanes with DPP only (row_shr 1/2/4/8 inside each row of 16, then row_bcast15 / row_bcast31
// carry the row totals): 6 VALU instructions, where six __shfl_up steps are six ds_bpermute round trips through the LDS unit
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {  // inclusive running maximum over the 64 lanes, DPP only (lane 63 = the wave's)
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}

// force a wave-uniform value into an SGPR (loop cursors derived from ballots stay scalar: SALU arithmetic and
// s_cbranch instead of VALU + exec-mask branches)
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

}  // namespace xpng
