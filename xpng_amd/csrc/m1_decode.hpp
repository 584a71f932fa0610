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
#include <string>
#include <vector>

#include "common.hpp"

namespace xpng {

struct DecTile {  // per-tile parse result (device)
    uint64_t off;       // byte offset of the tile blob in d_blobs
    uint32_t type;      // tile type byte (0 = raw rows)
    uint32_t kbytes;    // bytes of k words
    uint32_t blk_off[10];  // block offsets relative to the blob start
    uint32_t blk_n[10];    // symbols per block
    uint32_t ctx_start[10];  // start of context stream c inside the tile's symbol area (ctx_start[9] = total coded)
};

struct DecodeWs {
    uint64_t cap_tiles = 0, cap_plane = 0;
    DecTile *d_info = nullptr;
    uint64_t *d_off = nullptr;
    uint8_t *d_ctxsym = nullptr, *d_asym = nullptr, *d_alpha = nullptr, *d_nlseq = nullptr;
    uint32_t *d_resid = nullptr;
};
inline void decode_ws_free(DecodeWs &w) {
    void *p[] = {w.d_info, w.d_off, w.d_ctxsym, w.d_asym, w.d_alpha, w.d_nlseq, w.d_resid};
    for (void *q : p) if (q) (void)hipFree(q);
    w = DecodeWs();
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
__global__ void k_dec_parse(const uint8_t *__restrict__ blobs, const uint64_t *__restrict__ off, uint32_t cnt,
                            uint32_t spt, DecTile *__restrict__ info) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt) return;
    DecTile d{};
    d.off = off[j];
    const uint8_t *f = blobs + d.off;
    const uint32_t h0 = ld32u(f);
    d.type = h0 >> 24;
    if (d.type != 0) {
        const uint32_t ksz = ld32u(f + 4);  // includes itself (libxpng.c:556)
        d.kbytes = ksz - 4;
        uint32_t o = 4 + ksz, acc = 0;
        for (uint32_t c = 0; c < spt; c++) {
            const uint32_t b0 = ld32u(f + o), ty = b0 >> 24;
            d.blk_off[c] = o;
            d.blk_n[c] = ty == 0 ? 0 : (ld32u(f + o + 4) & 0xFFFFFF);
            if (c < 9) { d.ctx_start[c] = acc; acc += d.blk_n[c]; }
            o += ty == 0 ? 4 : (b0 & 0xFFFFFF);
        }
        d.ctx_start[9] = acc;
    }
    info[j] = d;
}

// --------------------------------------------------------------------------------------------------
// one v2 block -> symbols (decompress_block_v2, libxpng.c:429-493).  Single-wave workgroup.
// LDS: slot2sym[32768] bytes, fc[256] (F | cum<<16), F32/cum scratch.
__global__ __launch_bounds__(64) void k_rans2_decode(const uint8_t *__restrict__ blobs, const DecTile *__restrict__ info,
                                                     const TileDesc *__restrict__ tiles, uint32_t t0, uint32_t spt,
                                                     uint8_t *__restrict__ ctxsym, uint8_t *__restrict__ asym) {
    __shared__ uint8_t slot2sym[1 << 15];
    __shared__ uint32_t fc[256];
    __shared__ uint32_t Fs[260];
    const uint32_t j = blockIdx.x / spt, c = blockIdx.x % spt, lane = threadIdx.x & 63;
    const DecTile d = info[j];
    if (d.type == 0) return;
    const TileDesc t = tiles[t0 + j];
    const uint8_t *in = blobs + d.off + d.blk_off[c];
    uint8_t *out = c < 9 ? ctxsym + t.pbase + d.ctx_start[c] : asym + t.pbase;
    const uint32_t h0 = ld32u(in), type = h0 >> 24;
    if (type == 0) return;
    const uint32_t csz = h0 & 0xFFFFFF;
    const uint8_t *end = in + csz;
    const uint32_t h1 = ld32u(in + 4), n = h1 & 0xFFFFFF, v2 = h1 >> 24;
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
    const uint32_t h2 = ld32u(in + 8);
    const int pb = (int)(h2 >> 24);
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
    {   // cum by 4-per-lane partial sums + wave scan; fc[i] = F | cum << 16
        const uint32_t b = lane * 4;
        const uint32_t f0 = b + 0 < N ? Fs[b + 0] : 0, f1 = b + 1 < N ? Fs[b + 1] : 0, f2 = b + 2 < N ? Fs[b + 2] : 0, f3 = b + 3 < N ? Fs[b + 3] : 0;
        const uint32_t tot = f0 + f1 + f2 + f3;
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t v = __shfl_up(incl, dd);
            if ((int)lane >= dd) incl += v;
        }
        const uint32_t c0 = incl - tot, c1 = c0 + f0, c2 = c1 + f1, c3 = c2 + f2;
        if (b + 0 < N) fc[b + 0] = f0 | (c0 << 16);
        if (b + 1 < N) fc[b + 1] = f1 | (c1 << 16);
        if (b + 2 < N) fc[b + 2] = f2 | (c2 << 16);
        if (b + 3 < N) fc[b + 3] = f3 | (c3 << 16);
    }
    __syncthreads();
    {   // slot -> symbol: every lane fills slots by binary search over cum (N <= 256 -> 8 probes)
        const uint32_t scale = 1u << pb;
        for (uint32_t s = lane; s < scale; s += 64) {
            uint32_t lo = 0, hi = N - 1;  // largest i with cum[i] <= s and F[i] > 0 region containing s
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if ((fc[mid] >> 16) <= s) lo = mid; else hi = mid - 1;
            }
            // symbols with F == 0 share their cum with the next used one: step down to the one that owns the slot
            // (the used symbol is the LAST index among equal cums whose F > 0 ... the search above lands on the
            // largest index with cum <= s; zero-width followers have cum == next cum > s only if they sit after)
            while (lo > 0 && (fc[lo] & 0xFFFF) == 0) lo--;
            slot2sym[s] = (uint8_t)lo;
        }
    }
    __syncthreads();
    // ---- the recurrence, backwards (libxpng.c:467-489).  lane 0 = state0, lane 1 = state1.
    const uint32_t mask = (1u << pb) - 1;
    const uint8_t *sp = table - 16;  // state0 at table-16, state1 at table-8
    uint64_t s = lane < 2 ? ld64u(sp + 8 * lane) : RANS_L;
    int32_t rw = (int32_t)((sp - words) >> 2);  // words still unread below the states; both lanes track it
    int64_t i = (int64_t)n;
    if (n & 1) {  // odd tail comes from state0 only
        i--;
        if (lane == 0) {
            const uint32_t slot = (uint32_t)s & mask, sym = slot2sym[slot], e = fc[sym];
            out[i] = (uint8_t)sym;
            s = (uint64_t)(e & 0xFFFF) * (s >> pb) + slot - (e >> 16);
        }
        const uint32_t need0 = __shfl((lane == 0 && s < RANS_L) ? 1u : 0u, 0);
        if (need0) {
            if (rw > 0) rw--;
            if (lane == 0) s = (s << 32) | ld32u(words + 4 * (int64_t)rw);
        }
    }
    for (i -= 2; i >= 0; i -= 2) {
        // speculative reads of the next two words (addresses depend only on rw, not on this step's states)
        const int32_t r1 = rw > 0 ? rw - 1 : 0, r2 = rw > 1 ? rw - 2 : 0;
        const uint32_t w1 = ld32u(words + 4 * (int64_t)r1), w2 = ld32u(words + 4 * (int64_t)r2);
        uint32_t need = 0;
        if (lane < 2) {
            const uint32_t slot = (uint32_t)s & mask, sym = slot2sym[slot], e = fc[sym];
            out[i + lane] = (uint8_t)sym;
            s = (uint64_t)(e & 0xFFFF) * (s >> pb) + slot - (e >> 16);
            need = s < RANS_L ? 1u : 0u;
        }
        const uint32_t other = swap_pair(need);
        // state1 refills first (libxpng.c:486-487)
        if (lane == 1 && need) s = (s << 32) | w1;
        if (lane == 0 && need) s = (s << 32) | (other ? w2 : w1);
        rw -= (int32_t)(need + other);
        if (rw < 0) rw = 0;
    }
}

// --------------------------------------------------------------------------------------------------
// alpha plane (RGBA).  a(x,y) = a(left) + d, except column 0: a(0,y) = a(0,y-1) + d (libxpng.c:798-800 with
// pr = p1x_ for rows 0 / interior and p1y_ for column 0).  grid = tiles, block = 1024.
__global__ __launch_bounds__(1024) void k_dec_alpha(const uint8_t *__restrict__ blobs, const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, uint32_t t0,
                                                    const uint8_t *__restrict__ asym, uint8_t *__restrict__ alpha) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DecTile d = info[j];
    if (d.type == 0) return;
    const TileDesc t = tiles[t0 + j];
    const uint8_t *sy = asym + t.pbase;  // sy[i-1] = symbol of pixel i
    uint8_t *al = alpha + t.pbase;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    // first pixel's alpha: 4th byte of the first k word (MSB-first R,G,B,A)
    const uint32_t a0 = ld32u(blobs + d.off + 8) & 0xFF;
    // ---- column 0: inclusive scan down the rows
    if (tid == 0) s_carry = a0;
    __syncthreads();
    if (tid == 0) al[0] = (uint8_t)a0;
    for (uint32_t y0 = 1; y0 < t.h; y0 += 1024) {
        const uint32_t y = y0 + tid;
        const uint32_t dv = y < t.h ? (uint32_t)zz_dec(sy[(uint64_t)y * t.w - 1]) & 255u : 0u;
        uint32_t incl = dv;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t v = __shfl_up(incl, dd);
            if ((int)lane >= dd) incl += v;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        uint32_t base = s_carry;
        for (uint32_t w2 = 0; w2 < wv; w2++) base += s_wave[w2];
        if (y < t.h) al[(uint64_t)y * t.w] = (uint8_t)(base + incl);
        __syncthreads();
        if (tid == 1023) s_carry = (base + incl) & 255u;
        __syncthreads();
    }
    __syncthreads();
    // ---- rows: each wave scans whole rows left to right, 64 pixels per step
    for (uint32_t y = wv; y < t.h; y += 16) {
        uint32_t carry = al[(uint64_t)y * t.w];
        for (uint32_t x0 = 1; x0 < t.w; x0 += 64) {
            const uint32_t x = x0 + lane;
            const uint32_t dv = x < t.w ? (uint32_t)zz_dec(sy[(uint64_t)y * t.w + x - 1]) & 255u : 0u;
            uint32_t incl = dv;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t v = __shfl_up(incl, dd);
                if ((int)lane >= dd) incl += v;
            }
            if (x < t.w) al[(uint64_t)y * t.w + x] = (uint8_t)(carry + incl);
            carry = (carry + __shfl(incl, 63)) & 255u;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// context chain (libxpng.c:803: nl = *cx[nl]++, starting from 0).  One wave per tile; lane c < 9 owns queue c:
// an 8-symbol register window plus one prefetched 8-byte chunk.  Each step: broadcast the head of the current
// queue with v_readlane, pop it on the owning lane.  Output: nl sequence in coded-pixel order.
__global__ __launch_bounds__(64) void k_dec_walk(const DecTile *__restrict__ info, const TileDesc *__restrict__ tiles,
                                                 uint32_t t0, const uint8_t *__restrict__ ctxsym,
                                                 uint8_t *__restrict__ nlseq) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x & 63;
    const DecTile d = info[j];
    if (d.type == 0) return;
    const TileDesc t = tiles[t0 + j];
    const uint32_t total = d.ctx_start[9];
    const uint8_t *base = ctxsym + t.pbase;
    uint8_t *out = nlseq + t.pbase;
    // per-lane queue state (lanes >= 9 idle)
    const uint32_t qs = lane < 9 ? d.ctx_start[lane] : 0;
    uint64_t rd = qs;              // next byte index to load into the prefetch chunk
    uint64_t win = 0, nxt = 0;     // current window (low byte = head), prefetched chunk
    uint32_t have = 0, nhave = 0;  // valid bytes in win / nxt
    auto load_chunk = [&](uint64_t &dst, uint32_t &cnt) {  // up to 8 bytes starting at base[rd], byte-wise (unaligned queue starts)
        const uint64_t a = rd & ~7ull;
        const uint64_t raw = *reinterpret_cast<const uint64_t *>(base + a);  // planes are 256-B padded: always in bounds
        const uint32_t skip = (uint32_t)(rd - a);
        dst = raw >> (8 * skip);
        cnt = 8 - skip;
        rd = a + 8;
    };
    if (lane < 9) { load_chunk(win, have); load_chunk(nxt, nhave); }
    uint32_t cur = 0;
    uint64_t packed = 0;  // 8 output symbols staged before one 8-byte store
    for (uint32_t k = 0; k < total; k++) {
        const uint32_t head = (uint32_t)win & 0xFF;
        const uint32_t sym = __builtin_amdgcn_readlane((int)head, (int)cur);
        if (lane == cur) {
            win >>= 8;
            if (--have == 0) { win = nxt; have = nhave; load_chunk(nxt, nhave); }
        }
        cur = sym;
        packed |= (uint64_t)sym << (8 * (k & 7));
        if ((k & 7) == 7) { if (lane == 0) *reinterpret_cast<uint64_t *>(out + (k & ~7u)) = packed; packed = 0; }
    }
    if ((total & 7) && lane == 0) {
        for (uint32_t r = 0; r < (total & 7); r++) out[(total & ~7u) + r] = (uint8_t)(packed >> (8 * r));
    }
}

// --------------------------------------------------------------------------------------------------
// residual extraction.  Walks the tile's pixels in raster order, 1024 per step: coded flag (alpha != 0),
// coded index = running count, bit cursor = running sum of 3*nl; pulls 3*nl bits out of k, undoes zig-zag
// and the green subtraction, and stores one packed word per pixel: r | g<<8 | b<<16 | coded<<24.
template <int PXSZ>
__global__ __launch_bounds__(1024) void k_dec_resid(const uint8_t *__restrict__ blobs, const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, uint32_t t0,
                                                    const uint8_t *__restrict__ alpha, const uint8_t *__restrict__ nlseq,
                                                    uint32_t *__restrict__ resid) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DecTile d = info[j];
    if (d.type == 0) return;
    const TileDesc t = tiles[t0 + j];
    const uint8_t *kbase = blobs + d.off + 8, *kend = kbase + d.kbytes;
    const uint8_t *al = alpha + t.pbase, *nls = nlseq + t.pbase;
    uint32_t *rs = resid + t.pbase;
    const int useG = d.type & 1;
    __shared__ uint32_t s_wc[16], s_wb[16];
    uint32_t run_cnt = 0, run_bits = 8 * PXSZ;
    const uint64_t lt = lanemask_lt();
    for (uint32_t i0 = 0; i0 < t.n; i0 += 1024) {
        const uint32_t i = i0 + tid;
        bool coded = i < t.n && i > 0;
        if (PXSZ == 4 && coded) coded = al[i] != 0;
        const uint64_t m = __ballot(coded);
        const uint32_t rank = (uint32_t)__popcll(m & lt), wcnt = (uint32_t)__popcll(m);
        if (lane == 0) s_wc[wv] = wcnt;
        __syncthreads();
        uint32_t cbase = run_cnt, ctot = 0;
        for (uint32_t w2 = 0; w2 < 16; w2++) { const uint32_t v = s_wc[w2]; if (w2 < wv) cbase += v; ctot += v; }
        const uint32_t nl = coded ? nls[cbase + rank] : 0;
        const uint32_t len = 3 * nl;
        uint32_t incl = len;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t v = __shfl_up(incl, dd);
            if ((int)lane >= dd) incl += v;
        }
        if (lane == 63) s_wb[wv] = incl;
        __syncthreads();
        uint32_t bbase = run_bits, btot = 0;
        for (uint32_t w2 = 0; w2 < 16; w2++) { const uint32_t v = s_wb[w2]; if (w2 < wv) bbase += v; btot += v; }
        if (i < t.n) {
            uint32_t word = 0;
            if (coded) {
                int dr = 0, dg = 0, db = 0;
                if (nl) {
                    const uint32_t ob = bbase + incl - len;
                    const uint8_t *wp = kbase + (uint64_t)(ob >> 5) * 4;
                    const uint64_t two = ((uint64_t)(wp < kend ? ld32u(wp) : 0u) << 32) | (wp + 4 < kend ? ld32u(wp + 4) : 0u);
                    const uint32_t v = (uint32_t)((two >> (64 - (ob & 31) - len)) & ((1u << len) - 1));
                    const uint32_t mk = (1u << nl) - 1;
                    dr = zz_dec((int)(v >> (2 * nl))); dg = zz_dec((int)((v >> nl) & mk)); db = zz_dec((int)(v & mk));
                }
                const uint32_t y = i / t.w, x = i - y * t.w;
                if (useG && x > 0 && y > 0) { dr += dg; db += dg; }  // libxpng.c:813
                word = ((uint32_t)dr & 255u) | (((uint32_t)dg & 255u) << 8) | (((uint32_t)db & 255u) << 16) | (1u << 24);
            }
            rs[i] = word;
        }
        run_cnt += ctot; run_bits += btot;
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
// reconstruction.  Raw tiles: row copy (libxpng.c:846).  Coded tiles: pixel (x,y) needs reconstructed L, U, UL, so
// rows advance as an anti-diagonal wavefront: thread r handles row yb+r and, at step s, column s-r.  Its U is what
// thread r-1 produced one step earlier (LDS, double-buffered by step parity), its UL is its previous U, its L its own
// previous output.  Tiles taller than 1024 rows run in bands; a band's first row reads U from the raster.
template <int PXSZ>
__global__ __launch_bounds__(1024) void k_dec_recon(const uint8_t *__restrict__ blobs, const DecTile *__restrict__ info,
                                                    const TileDesc *__restrict__ tiles, uint32_t t0,
                                                    const uint8_t *__restrict__ alpha, const uint32_t *__restrict__ resid,
                                                    uint8_t *__restrict__ raster, uint64_t bpr) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const DecTile d = info[j];
    const TileDesc t = tiles[t0 + j];
    uint8_t *dst = raster + (uint64_t)t.y * bpr + (uint64_t)t.x * PXSZ;
    if (d.type == 0) {
        const uint8_t *src = blobs + d.off + 4;
        const uint64_t row = (uint64_t)t.w * PXSZ;
        for (uint64_t b = tid; b < row * t.h; b += 1024) { const uint64_t y = b / row, o = b - y * row; dst[y * bpr + o] = src[b]; }
        return;
    }
    const int useGrad = (d.type >> 1) & 1;
    const uint8_t *al = alpha + t.pbase;
    const uint32_t *rs = resid + t.pbase;
    __shared__ uint32_t s_row[2][1024];
    // first pixel from the head of k (libxpng.c:850): bytes MSB-first
    const uint32_t kw0 = ld32u(blobs + d.off + 8);
    uint32_t first = ((kw0 >> 24) & 255u) | (((kw0 >> 16) & 255u) << 8) | (((kw0 >> 8) & 255u) << 16);
    if (PXSZ == 4) first |= (kw0 & 255u) << 24;
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
                    const bool coded = (rw >> 24) & 1;
                    if (coded) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const int l = (L >> (8 * c)) & 255, u = (U >> (8 * c)) & 255, ul = (UL >> (8 * c)) & 255;
                            int pred;
                            if (y == 0) pred = l;
                            else if (x == 0) pred = u;
                            else pred = useGrad ? pred_grad(l, u, ul) : pred_avg(l, u);
                            outpx |= (((rw >> (8 * c)) + (uint32_t)pred) & 255u) << (8 * c);
                        }
                    }
                    if (PXSZ == 4) outpx |= (uint32_t)al[i] << 24;  // alpha==0 -> whole pixel 0 (libxpng.c:802)
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

// --------------------------------------------------------------------------------------------------
inline int decode_m1_launch(DecodeWs &ws, const std::vector<TileDesc> &tiles, const TileDesc *d_tiles, uint64_t W, int pxsz,
                            const uint8_t *d_blobs, uint64_t blobs_len, const uint64_t *tile_off, uint32_t t0, uint32_t t1,
                            uint8_t *d_raster, hipStream_t s, std::string &err) {
    const uint32_t cnt = t1 - t0, spt = pxsz == 4 ? 10 : 9;
    const uint64_t plane = tiles.back().pbase + rup(tiles.back().n + 8, 256);
    auto bad = [&](const char *m) { err = m; return 1; };
    if (ws.cap_tiles < tiles.size() || ws.cap_plane < plane) {
        decode_ws_free(ws);
        if (hipMalloc((void **)&ws.d_info, tiles.size() * sizeof(DecTile)) != hipSuccess || hipMalloc((void **)&ws.d_off, tiles.size() * 8) != hipSuccess ||
            hipMalloc((void **)&ws.d_ctxsym, plane + 64) != hipSuccess || hipMalloc((void **)&ws.d_asym, plane + 64) != hipSuccess ||
            hipMalloc((void **)&ws.d_alpha, plane + 64) != hipSuccess || hipMalloc((void **)&ws.d_nlseq, plane + 64) != hipSuccess ||
            hipMalloc((void **)&ws.d_resid, 4 * plane + 64) != hipSuccess)
            return bad("hipMalloc failed (decode workspace)");
        ws.cap_tiles = tiles.size(); ws.cap_plane = plane;
    }
    (void)blobs_len;
    if (hipMemcpyAsync(ws.d_off, tile_off, (uint64_t)cnt * 8, hipMemcpyHostToDevice, s) != hipSuccess) return bad("tile offset upload failed");
    const uint64_t bpr = W * (uint64_t)pxsz;
    k_dec_parse<<<(cnt + 63) / 64, 64, 0, s>>>(d_blobs, ws.d_off, cnt, spt, ws.d_info);
    k_rans2_decode<<<cnt * spt, 64, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, spt, ws.d_ctxsym, ws.d_asym);
    if (pxsz == 4) k_dec_alpha<<<cnt, 1024, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, ws.d_asym, ws.d_alpha);
    k_dec_walk<<<cnt, 64, 0, s>>>(ws.d_info, d_tiles, t0, ws.d_ctxsym, ws.d_nlseq);
    if (pxsz == 4) {
        k_dec_resid<4><<<cnt, 1024, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, ws.d_alpha, ws.d_nlseq, ws.d_resid);
        k_dec_recon<4><<<cnt, 1024, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, ws.d_alpha, ws.d_resid, d_raster, bpr);
    } else {
        k_dec_resid<3><<<cnt, 1024, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, ws.d_alpha, ws.d_nlseq, ws.d_resid);
        k_dec_recon<3><<<cnt, 1024, 0, s>>>(d_blobs, ws.d_info, d_tiles, t0, ws.d_alpha, ws.d_resid, d_raster, bpr);
    }
    if (hipGetLastError() != hipSuccess) return bad("decode kernel launch failed");
    return 0;
}

}  // namespace xpng
