/* xpng_oracle.h -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * A plain-C CPU restatement of the xPNG codec hot path (reference: /root/reference/libxpng.c),
 * written from the algorithm description in SURVEY.md §8(a) and pinned byte-for-byte against
 * the compiled reference (oracle/_ref, see oracle/Makefile and oracle/make_golden.py) and the
 * committed fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The shipped library (xpng_amd/csrc -> libxpng_hip.so) never does.
 *
 * The restatement is organised as explicit stages so that every HIP kernel has a CPU twin:
 *   tile table -> predictor chooser -> per-pixel symbol planes -> context/bit streams
 *   -> rANS blocks -> tile blob -> file container (and the mirror for decode).
 */
#ifndef XPNG_ORACLE_H
#define XPNG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XO_TILE_AREA (444u * 444u) /* libxpng.c:49 */
#define XO_NL_NONE 0xFFu           /* nl-plane marker: pixel emits no colour symbol */

typedef struct xo_tile {
    uint64_t x, y, w, h; /* top-left pixel and size inside the raster */
} xo_tile;

/* libxpng.c:51-83.  Returns the tile count; fills out[0..min(N,cap)). */
uint64_t xo_tile_table(uint64_t W, uint64_t H, int pxsz, xo_tile *out, uint64_t cap);

/* libxpng.c:688-721.  0 = ok.  *out == NULL means "unchanged"; otherwise a malloc'ed raster
 * with *alpha_out channels-3. */
int xo_normalize_rgba(const uint8_t *rgba, uint64_t w, uint64_t h, uint8_t **out, int *alpha_out);

/* libxpng.c:92-140 (pp_rgbx).  `pxsz` is the raster's pixel size; `as_rgb` mirrors mode 2's
 * call with PXSZ=3 (libxpng.c:663).  Also returns the four cost sums when sums != NULL. */
int xo_choose_predictor(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, uint32_t sums[4]);

/* Per-pixel symbol planes of one tile in tile-linear order (index = y*w + x), the arithmetic of
 * libxpng.c:497-519 with the routing stripped off.  Each plane has w*h entries.
 *   nl : 0..8 for a coded pixel, XO_NL_NONE for pixel 0 and for alpha==0 pixels
 *   r,g,b : zig-zag residual bytes (0 where nl is NONE)
 *   a : alpha symbol (RGBA only, may be NULL for RGB; entry 0 unused = 0) */
void xo_m1_planes(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, int pr,
                  uint8_t *nl, uint8_t *r, uint8_t *g, uint8_t *b, uint8_t *a);

/* Stream formation from planes (routing part of libxpng.c:500-508): nine context streams,
 * residual bit stream `k` (first pixel + 3*nl bits per coded pixel, MSB first), histograms. */
typedef struct xo_m1_streams {
    uint8_t *ctx[9];
    uint32_t ctx_n[9];
    uint32_t *kwords;
    uint32_t k_n;        /* number of u32 words */
    uint32_t F[9 * 16];  /* F[pl*16 + nl] */
    uint32_t FA[256];
} xo_m1_streams;
int xo_m1_form_streams(const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t,
                       const uint8_t *nl, const uint8_t *r, const uint8_t *g, const uint8_t *b,
                       const uint8_t *a, xo_m1_streams *out);
void xo_m1_streams_free(xo_m1_streams *s);

/* rANS v2 block (libxpng.c:307-427 / 429-493).  F is clobbered (normalised). Returns bytes. */
uint64_t xo_rans2_encode(uint32_t *F, unsigned nominalN, const uint8_t *in, uint64_t n, uint8_t *out, int pb);
uint64_t xo_rans2_decode(const uint8_t *in, uint8_t *out, uint64_t *n_out);

/* Whole tile.  `out` must hold w*h*pxsz + 4 + slack (xo_tile_blob_bound). Returns blob bytes, 0 on error. */
uint64_t xo_tile_blob_bound(const xo_tile *t, int pxsz);
uint64_t xo_encode_tile(int mode, const uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t, uint8_t *out);
int xo_decode_tile(int mode, const uint8_t *blob, uint8_t *raster, uint64_t W, int pxsz, const xo_tile *t);

/* File level, in memory (libxpng.c:723-789 / 963-997).  0 = ok.
 * threads: 0 = all online cores, clamped to the tile count (libxpng.c:147). */
int xo_encode_image(int level, const uint8_t *raster, uint64_t w, uint64_t h, int alpha,
                    uint8_t **file, uint64_t *file_len, int threads);
int xo_decode_image(const uint8_t *file, uint64_t file_len, uint8_t **raster,
                    uint64_t *w, uint64_t *h, int *alpha, int threads);

/* Same, but only the tile stage on an already-normalised raster: concatenated blobs in tile
 * order (what the GPU shim's xpnghip_encode_tiles returns).  *blobs is malloc'ed. */
int xo_encode_tiles(int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz,
                    uint8_t **blobs, uint64_t *blobs_len, int threads);
int xo_decode_tiles(int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h,
                    int pxsz, uint8_t *raster, int threads);

/* Wall-clock of the last xo_encode_image / xo_decode_image timed region in ns (validate -> all
 * tile blobs produced; after-input-in-memory -> all tiles decoded), as libxpng.c:727/760, 967/985. */
uint64_t xo_last_encode_ns(void);
uint64_t xo_last_decode_ns(void);

#ifdef __cplusplus
}
#endif
#endif
