/* xpng_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT (see xpng_oracle.h).
 *
 * CPU restatement of the xPNG hot path.  Parity status: PINNED -- byte-identical to the compiled
 * reference (oracle/_ref/xpng, built from /root/reference by oracle/Makefile) on the 17-image
 * corpus and on the synthetic edge set, levels 1/2/7, and to the fixtures in tests/golden/
 * (tests/test_oracle_golden.py).  Each function cites the reference lines it restates.
 */
#include "xpng_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ scalar helpers */

/* numBit, libxpng.c:19: 0 for 0, else 1+floor(log2 v). */
static inline int bit_width(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }

/* pix_toU, libxpng.c:20: wrap to int8, then zig-zag to 0..255. */
static inline int zz_enc(int d) {
    int v = (int8_t)d;
    return ((int)((unsigned)v << 1) ^ (v >> 31)) & 0xFF;
}
/* pix_toS, libxpng.c:21 */
static inline int zz_dec(int u) { return (u >> 1) ^ -(u & 1); }

/* libxpng.c:27 and :29 (arithmetic shift on the possibly negative gradient sum) */
static inline int pred_avg(int L, int U) { return (L + U + 1) >> 1; }
static inline int pred_grad(int L, int U, int UL) { return ((3 * L + 3 * U - 2 * UL) + 2) >> 2; }

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void wr32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void wr64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

/* ------------------------------------------------------------------ bit I/O (libxpng.c:8-17,145) */

typedef struct bitw {
    uint64_t acc;
    uint32_t pend;  /* bits waiting in acc */
    uint8_t *p;     /* next u32 slot (byte pointer; words are little-endian u32) */
} bitw;

static inline void bitw_put(bitw *w, unsigned c, uint64_t v) {
    w->acc = (w->acc << c) | v;
    w->pend += c;
    if (w->pend >= 32) { /* BITSTREAM_FLUSH */
        w->pend -= 32;
        wr32(w->p, (uint32_t)(w->acc >> w->pend));
        w->p += 4;
    }
}
/* Append without the flush test (first pixel of a tile: libxpng.c:547 writes 8*PXSZ bits raw) */
static inline void bitw_put_noflush(bitw *w, unsigned c, uint64_t v) {
    w->acc = (w->acc << c) | v;
    w->pend += c;
}
static inline void bitw_finish(bitw *w) { /* BITSTREAM_END */
    if (w->pend > 0) {
        wr32(w->p, (uint32_t)(w->acc << (32 - w->pend)));
        w->p += 4;
        w->pend = 0;
    }
}

typedef struct bitr {
    uint64_t acc;
    uint32_t have;
    const uint8_t *p, *end;
} bitr;

static inline uint32_t bitr_get(bitr *r, unsigned c) { /* FILL + READ */
    if (r->have < 32) {
        r->acc <<= 32;
        r->have += 32;
        if (r->p < r->end) { r->acc += rd32(r->p); r->p += 4; }
    }
    r->have -= c;
    return (uint32_t)((r->acc >> r->have) & ((1ull << c) - 1));
}

/* ------------------------------------------------------------------ tile table (libxpng.c:51-83) */

static void split_axis(uint64_t len, uint64_t base, uint64_t *count, uint64_t *first, uint64_t *second) {
    uint64_t rem = len % base;
    *count = len / base;
    *first = base + rem;
    *second = base;
    if (rem > base / 2) { /* remainder too big for one tile: split first tile in two */
        *count += 1;
        *second = *first / 2;
        *first = *second + (*first & 1);
    }
}

uint64_t xo_tile_table(uint64_t W, uint64_t H, int pxsz, xo_tile *out, uint64_t cap) {
    uint64_t nx, ny, w0, w1, bw, h0, h1, bh;
    if (W * H * (uint64_t)pxsz <= (uint64_t)XO_TILE_AREA * (uint64_t)pxsz) {
        nx = ny = 1; w0 = W; h0 = H; w1 = bw = h1 = bh = 0;
    } else {
        if (W < 444) { bw = W; bh = XO_TILE_AREA / W; }
        else if (H < 444) { bh = H; bw = XO_TILE_AREA / H; }
        else bw = bh = 444;
        split_axis(W, bw, &nx, &w0, &w1);
        split_axis(H, bh, &ny, &h0, &h1);
    }
    uint64_t n = nx * ny, k = 0;
    for (uint64_t j = 0; j < ny && out; j++) {
        uint64_t y = j == 0 ? 0 : (j == 1 ? h0 : h0 + h1 + (j - 2) * bh);
        uint64_t th = j == 0 ? h0 : (j == 1 ? h1 : bh);
        for (uint64_t i = 0; i < nx; i++, k++) {
            if (k >= cap) return n;
            out[k].x = i == 0 ? 0 : (i == 1 ? w0 : w0 + w1 + (i - 2) * bw);
            out[k].w = i == 0 ? w0 : (i == 1 ? w1 : bw);
            out[k].y = y;
            out[k].h = th;
        }
    }
    return n;
}

/* ------------------------------------------------------------------ normalize (libxpng.c:688-721) */

int xo_normalize_rgba(const uint8_t *rgba, uint64_t w, uint64_t h, uint8_t **out, int *alpha_out) {
    uint64_t n = w * h;
    int hidden_colour = 0, translucent = 0;
    *out = NULL;
    *alpha_out = 1;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *p = rgba + 4 * i;
        if (p[3] == 0 && (p[0] | p[1] | p[2])) { hidden_colour = 1; break; }
        if (p[3] != 255) translucent = 1;
    }
    if (hidden_colour) { /* zero RGB under alpha==0, stay RGBA */
        uint8_t *q = malloc(n * 4);
        if (!q) return 1;
        for (uint64_t i = 0; i < n; i++) {
            if (rgba[4 * i + 3]) memcpy(q + 4 * i, rgba + 4 * i, 4); else memset(q + 4 * i, 0, 4);
        }
        *out = q;
        return 0;
    }
    if (translucent) return 0;
    uint8_t *q = malloc(n * 3); /* fully opaque: drop alpha */
    if (!q) return 1;
    for (uint64_t i = 0; i < n; i++) { q[3 * i] = rgba[4 * i]; q[3 * i + 1] = rgba[4 * i + 1]; q[3 * i + 2] = rgba[4 * i + 2]; }
    *out = q;
    *alpha_out = 0;
    return 0;
}

/* ------------------------------------------------------------------ predictor chooser (libxpng.c:92-140) */

int xo_choose_predictor(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, uint32_t sums[4]) {
    uint32_t cost[4] = {0, 0, 0, 0};
    if (sums) memset(sums, 0, 16);
    if (t->w < 4 || t->h < 4) return 0; /* NB: drops the RGBA bit, libxpng.c:94 */
    const int64_t bpr = (int64_t)W * pxsz;
    const uint64_t xs = t->w / 4, ys = t->h / 4;
    for (uint64_t j = 0; j < ys; j++) {
        for (uint64_t i = 0; i < xs; i++) {
            const uint8_t *p = raster + (int64_t)(t->y + 4 * j + 3) * bpr + (int64_t)(t->x + 4 * i + 3) * pxsz;
            if (pxsz == 4 && p[3] == 0) continue;
            int d2[3], d3[3];
            for (int c = 0; c < 3; c++) {
                int L = p[c - pxsz], U = p[c - bpr], UL = p[c - bpr - pxsz];
                d2[c] = p[c] - pred_avg(L, U);
                d3[c] = p[c] - pred_grad(L, U, UL);
            }
            cost[0] += (uint32_t)bit_width((uint32_t)(zz_enc(d2[0]) | zz_enc(d2[1]) | zz_enc(d2[2])));
            cost[1] += (uint32_t)bit_width((uint32_t)(zz_enc(d2[0] - d2[1]) | zz_enc(d2[1]) | zz_enc(d2[2] - d2[1])));
            cost[2] += (uint32_t)bit_width((uint32_t)(zz_enc(d3[0]) | zz_enc(d3[1]) | zz_enc(d3[2])));
            cost[3] += (uint32_t)bit_width((uint32_t)(zz_enc(d3[0] - d3[1]) | zz_enc(d3[1]) | zz_enc(d3[2] - d3[1])));
        }
    }
    if (sums) memcpy(sums, cost, 16);
    int best = 0;
    for (int k = 1; k < 4; k++) if (cost[k] < cost[best]) best = k; /* first minimum wins */
    return (pxsz & 4) | best;
}

/* ------------------------------------------------------------------ mode-1 planes (libxpng.c:497-519) */

void xo_m1_planes(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, int pr,
                  uint8_t *nl, uint8_t *r, uint8_t *g, uint8_t *b, uint8_t *a) {
    const int64_t bpr = (int64_t)W * pxsz;
    const int useGrad = (pr >> 1) & 1, useG = pr & 1;
    const uint8_t *base = raster + (int64_t)t->y * bpr + (int64_t)t->x * pxsz;
    for (uint64_t y = 0; y < t->h; y++) {
        for (uint64_t x = 0; x < t->w; x++) {
            uint64_t i = y * t->w + x;
            const uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * pxsz;
            nl[i] = XO_NL_NONE; r[i] = g[i] = b[i] = 0;
            if (a) a[i] = 0;
            if (i == 0) continue; /* first pixel travels raw in the bit stream */
            if (pxsz == 4) {
                int pa = (y == 0 || x > 0) ? p[3 - pxsz] : p[3 - bpr]; /* libxpng.c:510-511: left, except column 0 */
                a[i] = (uint8_t)zz_enc(p[3] - pa);
                if (p[3] == 0) continue; /* libxpng.c:502 */
            }
            int d[3];
            for (int c = 0; c < 3; c++) {
                int pred;
                if (y == 0) pred = p[c - pxsz];
                else if (x == 0) pred = p[c - bpr];
                else {
                    int L = p[c - pxsz], U = p[c - bpr], UL = p[c - bpr - pxsz];
                    pred = useGrad ? pred_grad(L, U, UL) : pred_avg(L, U);
                }
                d[c] = p[c] - pred;
            }
            if (useG && y > 0 && x > 0) { d[0] -= d[1]; d[2] -= d[1]; } /* libxpng.c:513 */
            int zr = zz_enc(d[0]), zg = zz_enc(d[1]), zb = zz_enc(d[2]);
            r[i] = (uint8_t)zr; g[i] = (uint8_t)zg; b[i] = (uint8_t)zb;
            nl[i] = (uint8_t)bit_width((uint32_t)(zr | zg | zb));
        }
    }
}

/* ------------------------------------------------------------------ mode-1 stream formation (libxpng.c:500-508,547,556) */

void xo_m1_streams_free(xo_m1_streams *s) {
    if (!s) return;
    free(s->ctx[0]);
    free(s->kwords);
    memset(s, 0, sizeof *s);
}

int xo_m1_form_streams(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t,
                       const uint8_t *nl, const uint8_t *r, const uint8_t *g, const uint8_t *b,
                       const uint8_t *a, xo_m1_streams *out) {
    const uint64_t n = t->w * t->h;
    memset(out, 0, sizeof *out);
    uint8_t *pool = malloc(9 * (n + 1));
    uint32_t *kw = malloc(4 * (n * 3 / 4 + 8) + 16);
    if (!pool || !kw) { free(pool); free(kw); return 1; }
    for (int c = 0; c < 9; c++) out->ctx[c] = pool + (uint64_t)c * (n + 1);
    out->kwords = kw;
    bitw k = { 0, 0, (uint8_t *)kw };
    const uint8_t *p0 = raster + (int64_t)t->y * (int64_t)W * pxsz + (int64_t)t->x * pxsz;
    for (int c = 0; c < pxsz; c++) bitw_put_noflush(&k, 8, p0[c]);
    unsigned pl = 0;
    for (uint64_t i = 1; i < n; i++) {
        if (pxsz == 4) out->FA[a[i]]++;
        unsigned v = nl[i];
        if (v == XO_NL_NONE) continue;
        out->F[(pl << 4) + v]++;
        out->ctx[pl][out->ctx_n[pl]++] = (uint8_t)v;
        pl = v;
        if (v) bitw_put(&k, 3 * v, ((uint64_t)r[i] << (2 * v)) | ((uint64_t)g[i] << v) | b[i]);
    }
    bitw_finish(&k);
    out->k_n = (uint32_t)((k.p - (uint8_t *)kw) / 4);
    return 0;
}

/* ------------------------------------------------------------------ rANS common (libxpng.c:153-158) */

#define RANS_L (1ull << 31)

typedef struct rans_enc_sym {
    uint64_t rcp;
    uint32_t freq, bias, cmpl, rshift;
} rans_enc_sym;

/* Frequency normalisation to 2^pb with the sequential "steal" repair, libxpng.c:320-329 (=173-182).
 * cum has N+1 entries holding the raw cumulative counts on entry. */
static void normalise_freqs(uint32_t *F, uint32_t *cum, unsigned N, uint64_t total, int pb) {
    for (unsigned i = 1; i <= N; i++) cum[i] = (uint32_t)(((uint64_t)cum[i] << pb) / total);
    for (unsigned i = 0; i < N; i++) {
        if (!F[i] || cum[i + 1] != cum[i]) continue;
        uint32_t smallest = ~0u;
        unsigned donor = 0;
        for (unsigned j = 0; j < N; j++) {
            uint32_t f = cum[j + 1] - cum[j];
            if (f > 1 && f < smallest) { smallest = f; donor = j; }
        }
        if (donor < i) for (unsigned j = donor + 1; j <= i; j++) cum[j]--;
        else for (unsigned j = i + 1; j <= donor; j++) cum[j]++;
    }
    for (unsigned i = 0; i < N; i++) F[i] = cum[i + 1] - cum[i];
}

/* Encoder entry, libxpng.c:331-360 (=184-213): Alverson reciprocal, two 64-bit divisions. */
static void make_enc_syms(rans_enc_sym *e, const uint32_t *F, const uint32_t *cum, unsigned N, int pb) {
    for (unsigned i = 0; i < N; i++) {
        e[i].freq = F[i];
        e[i].cmpl = (1u << pb) - F[i];
        if (F[i] < 2) {
            e[i].rcp = ~0ull; e[i].rshift = 0; e[i].bias = cum[i] + ((1u << pb) - 1);
        } else {
            uint32_t sh = 0;
            while (F[i] > (1u << sh)) sh++;
            uint64_t hi_dividend = 1ull << (sh + 31);
            uint64_t q_hi = hi_dividend / F[i];
            uint64_t lo_dividend = (F[i] - 1) + ((hi_dividend % F[i]) << 32);
            uint64_t q_lo = lo_dividend / F[i];
            e[i].rcp = q_lo + (q_hi << 32);
            e[i].rshift = sh - 1;
            e[i].bias = cum[i];
        }
    }
}

static inline uint64_t rans_put(uint64_t s, const rans_enc_sym *e) { /* libxpng.c:375-376 */
    uint64_t q = (uint64_t)(((u128)s * e->rcp) >> 64) >> e->rshift;
    return s + e->bias + q * e->cmpl;
}
static inline uint64_t rans_limit(const rans_enc_sym *e, int pb) { return ((RANS_L >> pb) << 32) * e->freq; }

static uint64_t dword_aligned_bytes(uint64_t bits) { return (bits / 32) * 4 + (bits % 32 ? 4 : 0); } /* :303 */

/* ------------------------------------------------------------------ rANS v2 (mode 1) libxpng.c:307-427 */

uint64_t xo_rans2_encode(uint32_t *F, unsigned nominalN, const uint8_t *in, uint64_t n, uint8_t *out, int pb) {
    if (pb < 10 || pb > 15) return 0;
    if (n == 0) { wr32(out, 4); return 4; }
    int top = (int)nominalN - 1;
    while (top > 0 && F[top] == 0) top--;
    const unsigned N = (unsigned)top + 1, rawBits = (unsigned)bit_width((uint32_t)top);
    uint32_t cum[257], distinct = 0;
    cum[0] = 0;
    for (unsigned i = 0; i < N; i++) { cum[i + 1] = cum[i] + F[i]; distinct += F[i] != 0; }
    if (distinct == 1) {
        wr32(out, 8u | (1u << 24));
        wr32(out + 4, (uint32_t)n | ((uint32_t)in[0] << 24));
        return 8;
    }
    normalise_freqs(F, cum, N, n, pb);
    rans_enc_sym e[256];
    make_enc_syms(e, F, cum, N, pb);

    uint8_t *w = out + 12;
    uint64_t s0 = RANS_L, s1 = RANS_L;
    uint64_t i = 0;
    for (; i + 1 < n; i += 2) { /* forward, even symbol -> s0, odd -> s1; s0 spills first */
        const rans_enc_sym *a = &e[in[i]], *b = &e[in[i + 1]];
        if (s0 >= rans_limit(a, pb)) { wr32(w, (uint32_t)s0); w += 4; s0 >>= 32; }
        if (s1 >= rans_limit(b, pb)) { wr32(w, (uint32_t)s1); w += 4; s1 >>= 32; }
        s0 = rans_put(s0, a);
        s1 = rans_put(s1, b);
    }
    if (n & 1) {
        const rans_enc_sym *a = &e[in[i]];
        if (s0 >= rans_limit(a, pb)) { wr32(w, (uint32_t)s0); w += 4; s0 >>= 32; }
        s0 = rans_put(s0, a);
    }
    wr64(w, s0); wr64(w + 8, s1); w += 16;

    const uint32_t sparseBits = N + distinct * (uint32_t)pb;
    const int sparse = sparseBits < N * (uint32_t)pb;
    wr32(out + 4, (uint32_t)n | ((N - 2) << 24));
    wr32(out + 8, (uint32_t)((w - (out + 8)) / 4) | ((uint32_t)pb << 24));
    bitw tb = { 0, 0, w };
    for (unsigned k = 0; k < N; k++) {
        if (!sparse) bitw_put(&tb, (unsigned)pb, F[k]);
        else if (F[k]) bitw_put(&tb, (unsigned)pb + 1, F[k] + (1u << pb));
        else bitw_put(&tb, 1, 0);
    }
    bitw_finish(&tb);
    uint32_t csz = (uint32_t)(tb.p - out);
    wr32(out, csz | ((3u + (uint32_t)sparse) << 24));

    if (csz >= 8 + dword_aligned_bytes((uint64_t)rawBits * n)) { /* raw beats rANS: type 2 */
        bitw rb = { 0, 0, out + 8 };
        out[7] = (uint8_t)rawBits;
        for (uint64_t k = 0; k < n; k++) bitw_put(&rb, rawBits, in[k]);
        bitw_finish(&rb);
        csz = (uint32_t)(rb.p - out);
        wr32(out, csz | (2u << 24));
        /* out[4..7] keeps n | rawBits<<24 */
        return csz;
    }
    return csz;
}

/* libxpng.c:429-493 */
uint64_t xo_rans2_decode(const uint8_t *in, uint8_t *out, uint64_t *n_out) {
    uint32_t h0 = rd32(in), type = h0 >> 24;
    if (type == 0) { *n_out = 0; return 4; }
    const uint64_t csz = h0 & 0xFFFFFF;
    const uint8_t *end = in + csz;
    uint32_t h1 = rd32(in + 4), n = h1 & 0xFFFFFF, v2 = h1 >> 24;
    *n_out = n;
    if (type == 1) { memset(out, (int)v2, n); return csz; }
    if (type == 2) {
        bitr r = { 0, 0, in + 8, end };
        for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)bitr_get(&r, v2);
        return csz;
    }
    const unsigned N = v2 + 2;
    uint32_t h2 = rd32(in + 8);
    const int pb = (int)(h2 >> 24);
    const uint8_t *words = in + 12;                       /* first rANS word */
    const uint8_t *table = in + 8 + 4 * (uint64_t)(h2 & 0xFFFFFF);
    uint32_t F[258], cum[259];
    bitr tr = { 0, 0, table, end };
    for (unsigned i = 0; i < N; i++) {
        if (type == 3) F[i] = bitr_get(&tr, (unsigned)pb);
        else F[i] = bitr_get(&tr, 1) ? bitr_get(&tr, (unsigned)pb) : 0;
    }
    cum[0] = 0;
    for (unsigned i = 0; i < N; i++) cum[i + 1] = cum[i] + F[i];
    static __thread uint8_t slot2sym[1 << 15];
    for (unsigned i = 0; i < N; i++) if (F[i]) memset(slot2sym + cum[i], (int)i, F[i]);
    const uint64_t mask = (1ull << pb) - 1;
    const uint8_t *rp = table;
    uint64_t s1 = rd64(rp - 8), s0 = rd64(rp - 16);
    rp -= 16;
    int64_t i = (int64_t)n;
    if (n & 1) {
        unsigned s = slot2sym[s0 & mask];
        out[--i] = (uint8_t)s;
        s0 = F[s] * (s0 >> pb) + (s0 & mask) - cum[s];
        if (s0 < RANS_L) { if (rp > words) rp -= 4; s0 = (s0 << 32) | rd32(rp); }
    }
    for (i -= 2; i >= 0; i -= 2) {
        unsigned b = slot2sym[s1 & mask], a = slot2sym[s0 & mask];
        out[i + 1] = (uint8_t)b; out[i] = (uint8_t)a;
        s1 = F[b] * (s1 >> pb) + (s1 & mask) - cum[b];
        s0 = F[a] * (s0 >> pb) + (s0 & mask) - cum[a];
        if (s1 < RANS_L) { if (rp > words) rp -= 4; s1 = (s1 << 32) | rd32(rp); }
        if (s0 < RANS_L) { if (rp > words) rp -= 4; s0 = (s0 << 32) | rd32(rp); }
    }
    return csz;
}

/* ------------------------------------------------------------------ rANS v1 (mode 2) libxpng.c:160-301
 * The block is built downward from *top (bytes); symbols are st[0..n).  The frequency table or the
 * raw symbols go to the tile-wide shared bit stream `tb`.  Returns the new (lower) top. */
static uint8_t *rans1_encode(uint32_t *F, unsigned N, const uint8_t *st, uint64_t n, uint8_t *top, bitw *tb, int pb) {
    const unsigned rawBits = (unsigned)bit_width(N - 1);
    uint8_t *w = top;
    if (n == 0) { w -= 4; wr32(w, 4); return w; }
    uint32_t cum[257], distinct = 0;
    cum[0] = 0;
    for (unsigned i = 0; i < N; i++) { cum[i + 1] = cum[i] + F[i]; distinct += F[i] != 0; }
    if (distinct == 1) {
        w -= 4; wr32(w, (uint32_t)n + ((uint32_t)st[n - 1] << 24));
        w -= 4; wr32(w, 8u + (1u << 24));
        return w;
    }
    normalise_freqs(F, cum, N, n, pb);
    rans_enc_sym e[256];
    make_enc_syms(e, F, cum, N, pb);
    uint64_t s0 = RANS_L, s1 = RANS_L;
    int64_t i = (int64_t)n;
    if (n & 1) { i--; s0 = rans_put(s0, &e[st[i]]); } /* libxpng.c:218-225: no spill test on the odd tail */
    for (i -= 2; i >= 0; i -= 2) { /* backwards; s1 spills first */
        const rans_enc_sym *b = &e[st[i + 1]], *a = &e[st[i]];
        if (s1 >= rans_limit(b, pb)) { w -= 4; wr32(w, (uint32_t)s1); s1 >>= 32; }
        if (s0 >= rans_limit(a, pb)) { w -= 4; wr32(w, (uint32_t)s0); s0 >>= 32; }
        s1 = rans_put(s1, b);
        s0 = rans_put(s0, a);
    }
    w -= 8; wr64(w, s1);
    w -= 8; wr64(w, s0);
    uint32_t tabBits = (N - distinct) + distinct * ((uint32_t)pb + 1);
    const int sparse = tabBits < N * (uint32_t)pb;
    if (!sparse) tabBits = N * (uint32_t)pb;
    if ((uint64_t)tabBits + 8ull * (uint64_t)(top - w) >= (uint64_t)rawBits * n) { /* type 2: raw symbols into tb */
        for (uint64_t k = 0; k < n; k++) bitw_put(tb, rawBits, st[k]);
        w = top;
        w -= 4; wr32(w, (uint32_t)n);
        w -= 4; wr32(w, 8u + (2u << 24));
        return w;
    }
    for (unsigned k = 0; k < N; k++) {
        if (!sparse) bitw_put(tb, (unsigned)pb, F[k]);
        else if (F[k]) bitw_put(tb, (unsigned)pb + 1, F[k] + (1u << pb));
        else bitw_put(tb, 1, 0);
    }
    w -= 4; wr32(w, (uint32_t)n);
    w -= 4; wr32(w, (uint32_t)(top - w) + ((3u + (uint32_t)sparse) << 24));
    return w;
}

/* libxpng.c:262-301.  Returns block size in bytes (low 24 bits of the header). */
static uint32_t rans1_decode(const uint8_t *blk, unsigned N, uint8_t *x, bitr *tb, int pb) {
    const unsigned rawBits = (unsigned)bit_width(N - 1);
    uint32_t h0 = rd32(blk), size = h0 & 0xFFFFFF, type = h0 >> 24;
    const uint8_t *end = blk + size, *rp = blk + 4;
    if (type == 1) { uint32_t v = rd32(rp); memset(x, (int)(v >> 24), v & 0xFFFFFF); return size; }
    if (type == 2) { uint32_t n = rd32(rp); for (uint32_t i = 0; i < n; i++) x[i] = (uint8_t)bitr_get(tb, rawBits); return size; }
    if (type != 3 && type != 4) return size;
    uint32_t n = rd32(rp); rp += 4;
    uint32_t F[256], cum[257];
    for (unsigned i = 0; i < N; i++) {
        if (type == 3) F[i] = bitr_get(tb, (unsigned)pb);
        else F[i] = bitr_get(tb, 1) ? bitr_get(tb, (unsigned)pb) : 0;
    }
    cum[0] = 0;
    for (unsigned i = 0; i < N; i++) cum[i + 1] = cum[i] + F[i];
    static __thread uint8_t slot2sym[1 << 15];
    for (unsigned i = 0; i < N; i++) if (F[i]) memset(slot2sym + cum[i], (int)i, F[i]);
    if (rp + 16 > end) return size;
    const uint64_t mask = (1ull << pb) - 1;
    uint64_t s0 = rd64(rp), s1 = rd64(rp + 8);
    rp += 16;
    uint32_t i = 0;
    for (; i + 1 < n; i += 2) {
        unsigned a = slot2sym[s0 & mask], b = slot2sym[s1 & mask];
        x[i] = (uint8_t)a; x[i + 1] = (uint8_t)b;
        s0 = F[a] * (s0 >> pb) + (s0 & mask) - cum[a];
        s1 = F[b] * (s1 >> pb) + (s1 & mask) - cum[b];
        if (s0 < RANS_L) { s0 = (s0 << 32) | (rp + 4 <= end ? rd32(rp) : rd32(end - 4)); if (rp < end) rp += 4; }
        if (s1 < RANS_L) { s1 = (s1 << 32) | (rp + 4 <= end ? rd32(rp) : rd32(end - 4)); if (rp < end) rp += 4; }
    }
    if (n & 1) x[i] = slot2sym[s0 & mask];
    return size;
}

/* ------------------------------------------------------------------ tile: mode 1 (libxpng.c:534-571, 834-863) */

uint64_t xo_tile_blob_bound(const xo_tile *t, int pxsz) {
    uint64_t n = t->w * t->h;
    /* k words (<= 3n+4+4) + 9 ctx blocks (<= 12+4 each + 1.5 n total) + alpha (<= 2n + 1100) + headers; the
     * raw fallback caps the *kept* size at n*pxsz+4 but the attempt is built in full first. */
    return 8 + (3 * n + 16) + (2 * n + 9 * 64) + (pxsz == 4 ? 2 * n + 2048 : 0) + 4096;
}

static void copy_tile_rows_out(uint8_t *dst, const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t) {
    const uint64_t bpr = W * (uint64_t)pxsz, row = t->w * (uint64_t)pxsz;
    const uint8_t *src = raster + t->y * bpr + t->x * (uint64_t)pxsz;
    for (uint64_t y = 0; y < t->h; y++) memcpy(dst + y * row, src + y * bpr, row);
}
static void copy_tile_rows_in(uint8_t *raster, const uint8_t *src, uint64_t W, int pxsz, const xo_tile *t) {
    const uint64_t bpr = W * (uint64_t)pxsz, row = t->w * (uint64_t)pxsz;
    uint8_t *dst = raster + t->y * bpr + t->x * (uint64_t)pxsz;
    for (uint64_t y = 0; y < t->h; y++) memcpy(dst + y * bpr, src + y * row, row);
}

static uint64_t encode_tile_m1(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, uint8_t *out) {
    const uint64_t n = t->w * t->h;
    if (pxsz == 4 && (t->w < 4 || t->h < 4)) return 0; /* reference behaviour undefined */
    uint8_t *planes = malloc(5 * n);
    if (!planes) return 0;
    uint8_t *nl = planes, *r = nl + n, *g = r + n, *b = g + n, *a = pxsz == 4 ? b + n : NULL;
    int pr = xo_choose_predictor(raster, W, pxsz, t, NULL);
    xo_m1_planes(raster, W, pxsz, t, pr, nl, r, g, b, a);
    xo_m1_streams s;
    if (xo_m1_form_streams(raster, W, pxsz, t, nl, r, g, b, a, &s)) { free(planes); return 0; }
    uint8_t *f = out + 4;
    wr32(f, 4 + 4 * s.k_n);
    memcpy(f + 4, s.kwords, 4ull * s.k_n);
    f += 4 + 4ull * s.k_n;
    for (int c = 0; c < 9; c++) f += xo_rans2_encode(s.F + c * 16, 9, s.ctx[c], s.ctx_n[c], f, 12);
    if (pxsz == 4) f += xo_rans2_encode(s.FA, 256, a + 1, n - 1, f, 15);
    uint64_t fsz = (uint64_t)(f - out), raw = n * (uint64_t)pxsz + 4;
    if (fsz < raw) wr32(out, (1u << 28) + ((uint32_t)pr << 24) + (uint32_t)fsz);
    else { wr32(out, (uint32_t)raw); copy_tile_rows_out(out + 4, raster, W, pxsz, t); fsz = raw; }
    xo_m1_streams_free(&s);
    free(planes);
    return fsz;
}

static int decode_tile_m1(const uint8_t *blob, uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t) {
    const uint64_t n = t->w * t->h;
    const int64_t bpr = (int64_t)W * pxsz;
    uint32_t h0 = rd32(blob), type = h0 >> 24;
    if (type == 0) { copy_tile_rows_in(raster, blob + 4, W, pxsz, t); return 0; }
    const uint8_t *f = blob + 4;
    uint32_t ksz = rd32(f);
    bitr k = { 0, 0, f + 4, f + ksz };
    f += ksz;
    uint8_t *base = raster + (int64_t)t->y * bpr + (int64_t)t->x * pxsz;
    for (int c = 0; c < pxsz; c++) base[c] = (uint8_t)bitr_get(&k, 8);
    uint8_t *pool = malloc((pxsz == 4 ? 2 : 1) * (n + 16) + 16);
    if (!pool) return 1;
    const uint8_t *q[10];
    uint8_t *w = pool;
    for (int c = 0; c < 9; c++) { uint64_t m; f += xo_rans2_decode(f, w, &m); q[c] = w; w += m; }
    if (pxsz == 4) { uint64_t m; xo_rans2_decode(f, w, &m); q[9] = w; }
    const int useGrad = (type >> 1) & 1, useG = type & 1;
    unsigned cur = 0;
    for (uint64_t y = 0; y < t->h; y++) {
        for (uint64_t x = 0; x < t->w; x++) {
            if (!(x | y)) continue;
            uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * pxsz;
            if (pxsz == 4) {
                int pa = (y == 0 || x > 0) ? p[3 - pxsz] : p[3 - bpr];
                p[3] = (uint8_t)(zz_dec(*q[9]++) + pa);
                if (p[3] == 0) { p[0] = p[1] = p[2] = 0; continue; }
            }
            cur = *q[cur]++;
            int d[3] = {0, 0, 0};
            if (cur) {
                uint32_t v = bitr_get(&k, 3 * cur), m = (1u << cur) - 1;
                d[0] = (int)(v >> (2 * cur)); d[1] = (int)((v >> cur) & m); d[2] = (int)(v & m);
            }
            for (int c = 0; c < 3; c++) d[c] = zz_dec(d[c]);
            if (useG && y > 0 && x > 0) { d[0] += d[1]; d[2] += d[1]; }
            for (int c = 0; c < 3; c++) {
                int pred;
                if (y == 0) pred = p[c - pxsz];
                else if (x == 0) pred = p[c - bpr];
                else {
                    int L = p[c - pxsz], U = p[c - bpr], UL = p[c - bpr - pxsz];
                    pred = useGrad ? pred_grad(L, U, UL) : pred_avg(L, U);
                }
                p[c] = (uint8_t)(d[c] + pred);
            }
        }
    }
    free(pool);
    return 0;
}

/* ------------------------------------------------------------------ tile: mode 2 (libxpng.c:573-686, 865-961) */

static int tile_is_one_colour(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t) { /* :628-643 */
    const uint64_t bpr = W * (uint64_t)pxsz;
    const uint8_t *first = raster + t->y * bpr + t->x * (uint64_t)pxsz;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++)
            if (memcmp(first, first + y * bpr + x * (uint64_t)pxsz, (size_t)pxsz)) return 0;
    return 1;
}

static const unsigned class_alphabet[9] = {0, 8, 64, 8, 16, 32, 64, 128, 256}; /* libxpng.c:669 */

/* Gray tile: libxpng.c:583-626.  Returns blob size, or 0 if the tile is not gray. */
static uint64_t encode_tile_gray(const uint8_t *raster, uint64_t W, const xo_tile *t, uint8_t *out) {
    const uint64_t n = t->w * t->h;
    const int64_t bpr = (int64_t)W * 3;
    const uint8_t *base = raster + (int64_t)t->y * bpr + (int64_t)t->x * 3;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++) {
            const uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3;
            if (p[0] != p[1] || p[1] != p[2]) return 0;
        }
    uint8_t *sym = malloc(4 * n + 4 * (2 * n + 4096) + 4 * (n + 4096));
    if (!sym) return 0;
    uint8_t *st[4], *blk[4], *bits[4];
    for (int m = 0; m < 4; m++) {
        st[m] = sym + (uint64_t)m * n;
        blk[m] = sym + 4 * n + (uint64_t)m * (2 * n + 4096);
        bits[m] = sym + 4 * n + 4 * (2 * n + 4096) + (uint64_t)m * (n + 4096);
    }
    uint32_t F[4][256];
    memset(F, 0, sizeof F);
    uint64_t cnt = 0;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++) {
            if (!(x | y)) continue;
            const uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3;
            int v = p[0], s[4];
            if (y == 0) s[0] = s[1] = s[2] = s[3] = zz_enc(v - p[-3]);
            else if (x == 0) s[0] = s[1] = s[2] = s[3] = zz_enc(v - p[-bpr]);
            else {
                int L = p[-3], U = p[-bpr], UL = p[-bpr - 3];
                s[0] = zz_enc(v - L); s[1] = zz_enc(v - U);
                s[2] = zz_enc(v - pred_avg(L, U)); s[3] = zz_enc(v - pred_grad(L, U, UL));
            }
            for (int m = 0; m < 4; m++) { st[m][cnt] = (uint8_t)s[m]; F[m][s[m]]++; }
            cnt++;
        }
    uint64_t bestB = 1000000, bestR = 1000000; /* libxpng.c:606 */
    int best = 0;
    uint8_t *bestBlk = NULL;
    for (int m = 0; m < 4; m++) {
        bitw tb = { base[0], 8, bits[m] + 4 };
        uint8_t *top = blk[m] + (2 * n + 4096);
        uint8_t *lo = rans1_encode(F[m], 256, st[m], cnt, top, &tb, 15);
        bitw_finish(&tb);
        uint64_t bsz = (uint64_t)(tb.p - bits[m]), rsz = (uint64_t)(top - lo);
        wr32(bits[m], (uint32_t)bsz);
        if (bsz + rsz < bestB + bestR) { best = m; bestB = bsz; bestR = rsz; bestBlk = lo; }
    }
    uint64_t size;
    if (bestB + bestR >= n) { /* raw gray, type byte 0x28 */
        size = n + 4;
        wr32(out, (uint32_t)size + (5u << 27));
        for (uint64_t y = 0; y < t->h; y++)
            for (uint64_t x = 0; x < t->w; x++) out[4 + y * t->w + x] = base[(int64_t)y * bpr + (int64_t)x * 3];
    } else {
        size = bestB + bestR + 4;
        wr32(out, (uint32_t)size + (2u << 28) + ((uint32_t)best << 24));
        memcpy(out + 4, bits[best], bestB);
        memcpy(out + 4 + bestB, bestBlk, bestR);
    }
    free(sym);
    return size;
}

static uint64_t encode_tile_m2(const uint8_t *raster, uint64_t W, const xo_tile *t, uint8_t *out) {
    const uint64_t n = t->w * t->h;
    const int64_t bpr = (int64_t)W * 3;
    const uint8_t *base = raster + (int64_t)t->y * bpr + (int64_t)t->x * 3;
    if (tile_is_one_colour(raster, W, 3, t)) { /* libxpng.c:637-640 */
        wr32(out, (255u << 24) | 8);
        out[4] = base[0]; out[5] = base[1]; out[6] = base[2]; out[7] = 0;
        return 8;
    }
    uint64_t gsz = encode_tile_gray(raster, W, t, out);
    if (gsz) return gsz;

    /* colour tile: 9 context streams + 8 magnitude-class streams, libxpng.c:657-670 */
    const uint64_t ctxCap = n + 1, clsCap = 3 * n + 3;
    uint8_t *mem = malloc(9 * ctxCap + 8 * clsCap + (8 * n + 65536) + (4 * n + 65536));
    if (!mem) return 0;
    uint8_t *ctx[9], *cls[9];
    uint64_t ctxN[9] = {0}, clsN[9] = {0};
    for (int c = 0; c < 9; c++) ctx[c] = mem + (uint64_t)c * ctxCap;
    for (int c = 1; c < 9; c++) cls[c] = mem + 9 * ctxCap + (uint64_t)(c - 1) * clsCap;
    uint8_t *blkArea = mem + 9 * ctxCap + 8 * clsCap, *blkTop = blkArea + (8 * n + 65536);
    uint8_t *bitArea = blkTop;
    static __thread uint32_t F[9][256];
    memset(F, 0, sizeof F);
    int pr = xo_choose_predictor(raster, W, 3, t, NULL) & 3;
    const int useGrad = (pr >> 1) & 1, useG = pr & 1;
    unsigned pl = 0;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++) {
            if (!(x | y)) continue;
            const uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3;
            int d[3];
            for (int c = 0; c < 3; c++) {
                int pred;
                if (y == 0) pred = p[c - 3];
                else if (x == 0) pred = p[c - bpr];
                else { int L = p[c - 3], U = p[c - bpr], UL = p[c - bpr - 3]; pred = useGrad ? pred_grad(L, U, UL) : pred_avg(L, U); }
                d[c] = p[c] - pred;
            }
            if (useG && y > 0 && x > 0) { d[0] -= d[1]; d[2] -= d[1]; }
            int z[3] = { zz_enc(d[0]), zz_enc(d[1]), zz_enc(d[2]) };
            unsigned v = (unsigned)bit_width((uint32_t)(z[0] | z[1] | z[2]));
            F[0][(pl << 4) + v]++;
            ctx[pl][ctxN[pl]++] = (uint8_t)v;
            pl = v;
            if (v == 1) { unsigned s = (unsigned)((z[0] << 2) | (z[1] << 1) | z[2]); cls[1][clsN[1]++] = (uint8_t)s; F[1][s]++; }
            else if (v == 2) { unsigned s = (unsigned)((z[0] << 4) | (z[1] << 2) | z[2]); cls[2][clsN[2]++] = (uint8_t)s; F[2][s]++; }
            else if (v >= 3) for (int c = 0; c < 3; c++) { cls[v][clsN[v]++] = (uint8_t)z[c]; F[v][z[c]]++; }
        }
    bitw tb = { ((uint64_t)base[0] << 16) | ((uint64_t)base[1] << 8) | base[2], 24, bitArea + 4 };
    uint8_t *lo = blkTop, *ctxBlk[9], *clsBlk[9];
    for (int c = 0; c < 9; c++) { lo = rans1_encode(F[0] + c * 16, 9, ctx[c], ctxN[c], lo, &tb, 14); ctxBlk[c] = lo; }
    for (int c = 1; c < 9; c++) { lo = rans1_encode(F[c], class_alphabet[c], cls[c], clsN[c], lo, &tb, 14); clsBlk[c] = lo; }
    bitw_finish(&tb);
    uint64_t bsz = (uint64_t)(tb.p - bitArea), rsz = (uint64_t)(blkTop - lo), raw = n * 3;
    wr32(bitArea, (uint32_t)bsz);
    uint64_t size;
    if (bsz + rsz >= raw) { /* libxpng.c:675-677 */
        size = raw + 4;
        wr32(out, (uint32_t)size);
        copy_tile_rows_out(out + 4, raster, W, 3, t);
    } else {
        size = bsz + rsz + 4;
        wr32(out, (uint32_t)size + (1u << 28) + ((uint32_t)pr << 24));
        uint8_t *f = out + 4;
        memcpy(f, bitArea, bsz); f += bsz;
        for (int c = 0; c < 9; c++) { uint32_t sz = rd32(ctxBlk[c]) & 0xFFFFFF; memcpy(f, ctxBlk[c], sz); f += sz; }
        for (int c = 1; c < 9; c++) { uint32_t sz = rd32(clsBlk[c]) & 0xFFFFFF; memcpy(f, clsBlk[c], sz); f += sz; }
    }
    free(mem);
    return size;
}

static void fill_one_colour(uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, const uint8_t *px) { /* :916-927 */
    const uint64_t bpr = W * (uint64_t)pxsz;
    uint8_t *base = raster + t->y * bpr + t->x * (uint64_t)pxsz;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++) memcpy(base + y * bpr + x * (uint64_t)pxsz, px, (size_t)pxsz);
}

static int decode_tile_m2(const uint8_t *blob, uint8_t *raster, uint64_t W, const xo_tile *t) {
    const uint64_t n = t->w * t->h;
    const int64_t bpr = (int64_t)W * 3;
    uint32_t h0 = rd32(blob), type = h0 >> 24;
    uint8_t *base = raster + (int64_t)t->y * bpr + (int64_t)t->x * 3;
    if (type == 0) { copy_tile_rows_in(raster, blob + 4, W, 3, t); return 0; }
    if (type == 255) { fill_one_colour(raster, W, 3, t, blob + 4); return 0; }
    if ((type >> 4) == 2) { /* gray, libxpng.c:868-899 */
        const uint8_t *f = blob + 4;
        if (type & 8) {
            for (uint64_t y = 0; y < t->h; y++)
                for (uint64_t x = 0; x < t->w; x++) { uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3; p[0] = p[1] = p[2] = *f++; }
            return 0;
        }
        uint32_t bsz = rd32(f);
        bitr tb = { 0, 0, f + 4, f + bsz };
        f += bsz;
        base[0] = base[1] = base[2] = (uint8_t)bitr_get(&tb, 8);
        uint8_t *sym = malloc(n + 16);
        if (!sym) return 1;
        rans1_decode(f, 256, sym, &tb, 15);
        const uint8_t *s = sym;
        const unsigned m = type & 3;
        for (uint64_t y = 0; y < t->h; y++)
            for (uint64_t x = 0; x < t->w; x++) {
                if (!(x | y)) continue;
                uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3;
                int d = zz_dec(*s++), pred;
                if (y == 0) pred = p[-3];
                else if (x == 0) pred = p[-bpr];
                else {
                    int L = p[-3], U = p[-bpr], UL = p[-bpr - 3];
                    pred = m == 0 ? L : m == 1 ? U : m == 2 ? pred_avg(L, U) : pred_grad(L, U, UL);
                }
                p[0] = p[1] = p[2] = (uint8_t)(d + pred);
            }
        free(sym);
        return 0;
    }
    /* colour tile, libxpng.c:944-958 */
    const uint8_t *f = blob + 4;
    uint32_t bsz = rd32(f);
    bitr tb = { 0, 0, f + 4, f + bsz };
    f += bsz;
    for (int c = 0; c < 3; c++) base[c] = (uint8_t)bitr_get(&tb, 8);
    const uint64_t ctxCap = n + 16, clsCap = 3 * n + 16;
    uint8_t *mem = malloc(9 * ctxCap + 8 * clsCap);
    if (!mem) return 1;
    const uint8_t *ctx[9], *cls[9];
    for (int c = 0; c < 9; c++) { uint8_t *w = mem + (uint64_t)c * ctxCap; f += rans1_decode(f, 9, w, &tb, 14); ctx[c] = w; }
    for (int c = 1; c < 9; c++) { uint8_t *w = mem + 9 * ctxCap + (uint64_t)(c - 1) * clsCap; f += rans1_decode(f, class_alphabet[c], w, &tb, 14); cls[c] = w; }
    const int useGrad = (type >> 1) & 1, useG = type & 1;
    unsigned cur = 0;
    for (uint64_t y = 0; y < t->h; y++)
        for (uint64_t x = 0; x < t->w; x++) {
            if (!(x | y)) continue;
            uint8_t *p = base + (int64_t)y * bpr + (int64_t)x * 3;
            cur = *ctx[cur]++;
            int d[3] = {0, 0, 0};
            if (cur == 1) { unsigned s = *cls[1]++; d[0] = (int)(s >> 2); d[1] = (int)((s >> 1) & 1); d[2] = (int)(s & 1); }
            else if (cur == 2) { unsigned s = *cls[2]++; d[0] = (int)(s >> 4); d[1] = (int)((s >> 2) & 3); d[2] = (int)(s & 3); }
            else if (cur >= 3) { d[0] = *cls[cur]++; d[1] = *cls[cur]++; d[2] = *cls[cur]++; }
            for (int c = 0; c < 3; c++) d[c] = zz_dec(d[c]);
            if (useG && y > 0 && x > 0) { d[0] += d[1]; d[2] += d[1]; }
            for (int c = 0; c < 3; c++) {
                int pred;
                if (y == 0) pred = p[c - 3];
                else if (x == 0) pred = p[c - bpr];
                else { int L = p[c - 3], U = p[c - bpr], UL = p[c - bpr - 3]; pred = useGrad ? pred_grad(L, U, UL) : pred_avg(L, U); }
                p[c] = (uint8_t)(d[c] + pred);
            }
        }
    free(mem);
    return 0;
}

uint64_t xo_encode_tile(int mode, const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, uint8_t *out) {
    if (mode == 1) return encode_tile_m1(raster, W, pxsz, t, out);
    if (mode == 2 && pxsz == 3) return encode_tile_m2(raster, W, t, out);
    return 0;
}
int xo_decode_tile(int mode, const uint8_t *blob, uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t) {
    if (mode == 1) return decode_tile_m1(blob, raster, W, pxsz, t);
    if (mode == 2 && pxsz == 3) return decode_tile_m2(blob, raster, W, t);
    return 1;
}

/* ------------------------------------------------------------------ tile-parallel drivers (libxpng.c:146-151, until_fork/4_letters.c) */

typedef struct job {
    int mode, pxsz, decode, fail;
    const uint8_t *src;  /* encode: raster; decode: blobs */
    uint8_t *dst;        /* decode: raster */
    uint64_t W;
    const xo_tile *tiles;
    uint64_t ntiles;
    uint8_t **blob;      /* encode: per-tile malloc'ed blob */
    uint64_t *blob_len;  /* encode: out; decode: offsets */
    uint64_t next;
    pthread_mutex_t mu;
} job;

static void *worker(void *arg) {
    job *j = arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        uint64_t i = j->next < j->ntiles ? j->next++ : UINT64_MAX;
        pthread_mutex_unlock(&j->mu);
        if (i == UINT64_MAX) break;
        const xo_tile *t = &j->tiles[i];
        if (j->decode) {
            if (xo_decode_tile(j->mode, j->src + j->blob_len[i], j->dst, j->W, j->pxsz, t)) j->fail = 1;
        } else {
            uint8_t *buf = malloc(xo_tile_blob_bound(t, j->pxsz));
            uint64_t sz = buf ? xo_encode_tile(j->mode, j->src, j->W, j->pxsz, t, buf) : 0;
            if (!sz) { free(buf); j->fail = 1; j->blob[i] = NULL; j->blob_len[i] = 0; continue; }
            j->blob[i] = realloc(buf, sz);
            j->blob_len[i] = sz;
        }
    }
    return NULL;
}

static int run_job(job *j, int threads) {
    long T = threads > 0 ? threads : sysconf(_SC_NPROCESSORS_ONLN);
    if (T < 1) T = 1;
    if ((uint64_t)T > j->ntiles) T = (long)j->ntiles;
    pthread_mutex_init(&j->mu, NULL);
    pthread_t th[T];
    long started = 0;
    for (; started < T; started++) if (pthread_create(&th[started], NULL, worker, j)) { j->fail = 1; break; }
    for (long i = 0; i < started; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j->mu);
    return j->fail;
}

int xo_encode_tiles(int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz,
                    uint8_t **blobs, uint64_t *blobs_len, int threads) {
    uint64_t N = xo_tile_table(w, h, pxsz, NULL, 0);
    xo_tile *tiles = malloc(N * sizeof *tiles);
    uint8_t **blob = calloc(N, sizeof *blob);
    uint64_t *len = calloc(N, sizeof *len);
    if (!tiles || !blob || !len) return 1;
    xo_tile_table(w, h, pxsz, tiles, N);
    job j = { .mode = mode, .pxsz = pxsz, .src = raster, .W = w, .tiles = tiles, .ntiles = N, .blob = blob, .blob_len = len };
    int rc = run_job(&j, threads);
    uint64_t total = 0;
    for (uint64_t i = 0; i < N; i++) total += len[i];
    uint8_t *cat = rc ? NULL : malloc(total ? total : 1);
    if (cat) { uint64_t o = 0; for (uint64_t i = 0; i < N; i++) { memcpy(cat + o, blob[i], len[i]); o += len[i]; } }
    for (uint64_t i = 0; i < N; i++) free(blob[i]);
    free(blob); free(len); free(tiles);
    if (!cat) return 1;
    *blobs = cat; *blobs_len = total;
    return 0;
}

int xo_decode_tiles(int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h,
                    int pxsz, uint8_t *raster, int threads) {
    uint64_t N = xo_tile_table(w, h, pxsz, NULL, 0);
    xo_tile *tiles = malloc(N * sizeof *tiles);
    uint64_t *off = calloc(N, sizeof *off);
    if (!tiles || !off) return 1;
    xo_tile_table(w, h, pxsz, tiles, N);
    uint64_t o = 0;
    for (uint64_t i = 0; i < N; i++) { /* serial size walk, libxpng.c:982 */
        if (o + 4 > blobs_len) { free(tiles); free(off); return 1; }
        off[i] = o; o += rd32(blobs + o) & 0xFFFFFF;
    }
    job j = { .mode = mode, .pxsz = pxsz, .decode = 1, .src = blobs, .dst = raster, .W = w, .tiles = tiles, .ntiles = N, .blob_len = off };
    int rc = run_job(&j, threads);
    free(tiles); free(off);
    return rc;
}

/* ------------------------------------------------------------------ file container (libxpng.c:723-789, 963-997) */

static uint64_t g_enc_ns, g_dec_ns;
uint64_t xo_last_encode_ns(void) { return g_enc_ns; }
uint64_t xo_last_decode_ns(void) { return g_dec_ns; }
static uint64_t now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }

static int emit_level7(const uint8_t *raster, uint64_t w, uint64_t h, int alpha, uint8_t **file, uint64_t *len) {
    uint64_t s = w * h * (uint64_t)(3 + alpha);
    uint8_t *f = malloc(8 + s);
    if (!f) return 1;
    wr32(f, (uint32_t)(w - 1) | (7u << 24));
    wr32(f + 4, (uint32_t)(h - 1) | ((uint32_t)alpha << 24));
    memcpy(f + 8, raster, s);
    *file = f; *len = 8 + s;
    return 0;
}

int xo_encode_image(int level, const uint8_t *raster_in, uint64_t w, uint64_t h, int alpha,
                    uint8_t **file, uint64_t *file_len, int threads) {
    uint64_t t0 = now_ns();
    if (!raster_in || !w || !h || w > (1u << 24) || h > (1u << 24)) return 1;
    if (!(level == 1 || level == 2 || level == 7)) return 1;
    uint8_t *owned = NULL;
    const uint8_t *raster = raster_in;
    if (alpha) {
        if (xo_normalize_rgba(raster_in, w, h, &owned, &alpha)) return 1;
        if (owned) raster = owned;
    }
    const int pxsz = 3 + alpha;
    const uint64_t s = w * h * (uint64_t)pxsz;
    int rc = 1;
    if (s <= 4) level = 7; /* libxpng.c:735 */
    if (level == 7) { rc = emit_level7(raster, w, h, alpha, file, file_len); goto done; }
    if (level == 2) { /* whole-image single colour, libxpng.c:741-753 (checked before the RGBA fallback) */
        xo_tile whole = { 0, 0, w, h };
        if (tile_is_one_colour(raster, w, pxsz, &whole)) {
            uint8_t *f = malloc(8 + (size_t)pxsz);
            if (!f) goto done;
            wr32(f, (uint32_t)(w - 1) | (2u << 24));
            wr32(f + 4, (uint32_t)(h - 1) | ((uint32_t)alpha << 24) | (2u << 24));
            memcpy(f + 8, raster, (size_t)pxsz);
            *file = f; *file_len = 8 + (uint64_t)pxsz;
            rc = 0; goto done;
        }
    }
    if (alpha && level == 2) level = 1; /* libxpng.c:755 */
    if (alpha) { /* RGBA tiles narrower than 4 px: undefined in the reference; we store level 7 */
        xo_tile t0_;
        xo_tile_table(w, h, pxsz, &t0_, 1);
        if (w < 4 || h < 4 || t0_.w < 4 || t0_.h < 4) { rc = emit_level7(raster, w, h, alpha, file, file_len); goto done; }
    }
    {
        uint8_t *blobs; uint64_t blen;
        if (xo_encode_tiles(level, raster, w, h, pxsz, &blobs, &blen, threads)) goto done;
        g_enc_ns = now_ns() - t0;
        if (blen >= s) { free(blobs); rc = emit_level7(raster, w, h, alpha, file, file_len); goto done; } /* :771-777 */
        uint8_t *f = malloc(8 + blen);
        if (!f) { free(blobs); goto done; }
        wr32(f, (uint32_t)(w - 1) | ((uint32_t)level << 24));
        wr32(f + 4, (uint32_t)(h - 1) | ((uint32_t)alpha << 24));
        memcpy(f + 8, blobs, blen);
        free(blobs);
        *file = f; *file_len = 8 + blen;
        rc = 0;
    }
done:
    free(owned);
    return rc;
}

int xo_decode_image(const uint8_t *file, uint64_t file_len, uint8_t **raster_out,
                    uint64_t *w_out, uint64_t *h_out, int *alpha_out, int threads) {
    uint64_t t0 = now_ns();
    if (file_len < 8) return 1;
    uint32_t h0 = rd32(file), h1 = rd32(file + 4);
    uint64_t w = (h0 & 0xFFFFFF) + 1, h = (h1 & 0xFFFFFF) + 1;
    int alpha = (h1 >> 24) & 1, level = (int)(h0 >> 24);
    if (!(level == 1 || level == 2 || level == 7)) return 1;
    const int pxsz = 3 + alpha;
    uint64_t s = w * h * (uint64_t)pxsz;
    uint8_t *raster = malloc(s);
    if (!raster) return 1;
    *w_out = w; *h_out = h; *alpha_out = alpha; *raster_out = raster;
    if (level == 7) { if (file_len < 8 + s) { free(raster); return 1; } memcpy(raster, file + 8, s); return 0; }
    if (file_len == 11 + (uint64_t)alpha && (file[7] & 2)) { /* libxpng.c:976-980 */
        xo_tile whole = { 0, 0, w, h };
        fill_one_colour(raster, w, pxsz, &whole, file + 8);
        return 0;
    }
    int rc = xo_decode_tiles(level, file + 8, file_len - 8, w, h, pxsz, raster, threads);
    g_dec_ns = now_ns() - t0;
    if (rc) { free(raster); *raster_out = NULL; }
    return rc;
}
