/* xpng_hip.h -- C-ABI of libxpng_hip.so: the MI355X tile codec behind xpng_store / xpng_load.
 *
 * This is the "inner boundary" of SURVEY.md §8(b): it replaces the two pthread fan-outs of the
 * reference driver,
 *     spawn_and_wait(T, &d, 0, enc_1_th | enc_2_th)   reference libxpng.c:758
 *     spawn_and_wait(T, &d, 0, dec_1_th | dec_2_th)   reference libxpng.c:983
 * i.e. "given the raster and the tile table, produce every tile's blob" and the reverse.  Plain
 * pointers and sizes only; no C++ or torch types.  All functions return 0 on success, non-zero on
 * failure (the reference's _Bool convention, libxpng.c:729-731); xpnghip_last_error() describes the
 * last failure of the calling thread.  There is NO CPU fallback: without a usable HIP device every
 * compute entry point fails.
 *
 * Threading: like the reference (no globals; every call spawns and joins its own workers, until_fork/4_letters.c:9-17) the
 * host-buffer and staged-image entry points are RE-ENTRANT: any number of host threads may call them at the same time; each
 * call works on a context, a stream and staging buffers of its own (taken from a pool of idle ones and given back), and no
 * lock is held while a call runs.  A device-resident context (xpnghip_ctx) is a single-queue object owned by its caller.
 *
 * Environment read by the release library (each selects between forms that produce the same bytes):
 * XPNG_DEVICE, XPNG_GPUS, XPNG_WIDE_RANS, XPNG_NARROW_RANS, XPNG_SPLIT, XPNG_NO_SPLIT.  Switches that
 * exist for timing studies (kernel knock-outs, LDS pads, stamps, the wave probe, fake devices) are compiled only into
 * libxpng_hip_probes.so (`make probes`).
 */
#ifndef XPNG_HIP_H
#define XPNG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XPNGHIP_ABI_VERSION 2   /* 2: the staged image is a handle (re-entrant xpng_store); T == 0 means one device */

int xpnghip_abi_version(void);
int xpnghip_device_count(void);          /* visible HIP devices; 0 when none / no runtime */
const char *xpnghip_last_error(void);

/* ---- host-buffer entry points: what the host C driver (xpng_store_T / xpng_load_T) calls ----------
 *
 * xpnghip_encode_tiles  <->  libxpng.c:758 + the concatenation loop libxpng.c:764-769.
 *   raster: normalised interleaved raster, w*h*pxsz bytes (pxsz 3 or 4), host memory.
 *   mode:   1 (enc_1_th, RGB and RGBA) or 2 (enc_2_th, RGB only).
 *   *blobs: malloc()ed concatenation of all tile blobs in tile order (caller frees with free()).
 * xpnghip_decode_tiles  <->  libxpng.c:982-983.
 *   blobs:  the file body after the 8-byte header; tile sizes are walked serially as the reference does.
 *   raster: caller-allocated w*h*pxsz bytes, filled completely.
 */
int xpnghip_encode_tiles(int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz,
                         uint8_t **blobs, uint64_t *blobs_len);
int xpnghip_decode_tiles(int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h,
                         int pxsz, uint8_t *raster);
/* The same with the reference's worker count T (libxpng.c:146-151: T workers share the tile cursor, T = min(T, N)).  Here a
 * worker is a DEVICE of this process: T >= 1 uses min(T, visible devices, N) of them.  T == 0 (the reference's "auto") uses
 * ONE device: the multi-device path has not yet passed a byte-parity run on real peer devices, so it is opt-in (T > 1, or
 * XPNG_GPUS=<n> for T == 0).  Every device codes one contiguous, pixel-weighted tile range from its own band of the raster;
 * for the encode the blob ranges are gathered on the first device by peer copies (xGMI; peer access is queried and enabled
 * once per device pair) for the concatenation of libxpng.c:764-769, then copied to the host once.  When a pair of devices has
 * no peer access the copies are staged through host memory by the runtime: the call still succeeds and xpnghip_last_error()
 * then holds a note that starts with "note:".  The bytes do not depend on T (tiles are coded independently).
 * xpnghip_encode_tiles / xpnghip_decode_tiles are T = 1.  XPNG_DEVICE=<n> selects the first device (default 0); the
 * caller's current HIP device is restored before returning. */
int xpnghip_encode_tiles_T(uint64_t T, int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz,
                           uint8_t **blobs, uint64_t *blobs_len);
int xpnghip_decode_tiles_T(uint64_t T, int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h,
                           int pxsz, uint8_t *raster);
/* number of devices such a call would use for a w x h image (0 = no usable device) */
int xpnghip_devices_for(uint64_t T, uint64_t w, uint64_t h);
/* host-only (needs no device): the contiguous tile ranges a call on D devices would use; ranges[2k], ranges[2k+1] = [r0, r1) of
 * device k.  Returns the number of ranges (= min(D, tiles)), -1 on bad arguments or cap too small. */
int xpnghip_shard_ranges(uint64_t w, uint64_t h, int D, uint64_t *ranges, int cap);
/* Optional: gives every pooled idle object (contexts with their workspaces, staging buffers) back to the runtime.  Must not
 * run concurrently with other calls into this library. */
void xpnghip_shutdown(void);

/* ---- staged image: upload once, normalise and test on the device ------------------------------------
 *
 * xpng_store's pre-passes over the whole raster, moved off the host:
 *   xpnghip_image_begin          <->  normalize_RGBA, libxpng.c:688-721 (called at libxpng.c:733).  Uploads the caller's
 *                                     raster (w*h*pxsz_in bytes, host memory) and, for RGBA, applies the rule on the device:
 *                                     hidden colours under alpha 0 -> those pixels zeroed; no translucent pixel -> repacked
 *                                     to RGB.  *pxsz_out = 3 or 4 = bytes per pixel of the staged (normalised) raster.
 *   xpnghip_image_single_colour  <->  the whole-image test of libxpng.c:741-753: *single = 1 if every pixel equals the first.
 *   xpnghip_image_encode         <->  libxpng.c:758-769 on the staged raster (as xpnghip_encode_tiles).
 *   xpnghip_image_fetch               staged raster -> host (w*h*pxsz_out bytes): the level-7 and single-colour outputs.
 *   xpnghip_image_end                 hands the staging object back (always call it once begin has returned 0).
 * Every xpng_store call in flight stages its own image: begin returns a handle, the other calls take it.  Handles of different
 * threads are independent (the reference is re-entrant); one handle is used by one thread at a time.
 * xpnghip_normalize_device is the same rule for a caller whose RGBA raster already lives in HBM (on the caller's current
 * device): *rewritten = 0 means the raster is already normal (use d_rgba), 1 means d_out (npx * *pxsz_out bytes,
 * caller-allocated npx*4) holds it.  It synchronises `stream` once (the two flags come back to the host). */
typedef struct xpnghip_image xpnghip_image;
int xpnghip_image_begin(xpnghip_image **img, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz_in, int *pxsz_out);
int xpnghip_image_single_colour(xpnghip_image *img, int *single);
int xpnghip_image_encode(xpnghip_image *img, int mode, uint8_t **blobs, uint64_t *blobs_len);
int xpnghip_image_encode_T(xpnghip_image *img, uint64_t T, int mode, uint8_t **blobs, uint64_t *blobs_len);  /* T devices, as xpnghip_encode_tiles_T */
int xpnghip_image_fetch(xpnghip_image *img, uint8_t *dst);
void xpnghip_image_end(xpnghip_image *img);
int xpnghip_normalize_device(const void *d_rgba, uint64_t npx, void *d_out, int *pxsz_out, int *rewritten, void *stream);

/* ---- device-resident entry points (bench, multi-GPU sharding, pipelines) ---------------------------
 *
 * A context owns the tile table (libxpng.c:51-83) and every intermediate buffer for one raster
 * geometry on one device, so the hot path itself performs no allocation and no host sync except the
 * final 8-byte length read-back.  `stream` is a hipStream_t passed as void* (NULL = the context's own).
 * A context is a single-queue object: the calls made on one context must be ordered (issue them on one stream).
 * Its buffers are reused from call to call, and the decode workspace lives inside the encode stream scratch.
 * Use one context per pipeline slot to overlap work.
 */
typedef struct xpnghip_ctx xpnghip_ctx;

int xpnghip_ctx_create(xpnghip_ctx **ctx, int device, uint64_t w, uint64_t h, int pxsz);
/* Same, sized for up to `batch` rasters of this geometry per launch.  The entropy stage is one serial chain per
 * (tile, stream), so a single 4096^2 image (81 tiles) cannot fill 256 CUs; a batched launch runs the chains of all
 * images side by side (virtual tile = image * N + tile) at the latency of one image. */
int xpnghip_ctx_create_batch(xpnghip_ctx **ctx, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch);
/* Same, with workspace for tiles [r0, r1) only: a rank that codes one tile range of a large raster (multi-GPU sharding)
 * pays HBM for its share, not for the whole image.  Encode / decode calls must stay inside [r0, r1). */
int xpnghip_ctx_create_range(xpnghip_ctx **ctx, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch,
                             uint64_t r0, uint64_t r1);
uint32_t xpnghip_ctx_batch(const xpnghip_ctx *ctx);
void xpnghip_ctx_destroy(xpnghip_ctx *ctx);
uint64_t xpnghip_ctx_tile_count(const xpnghip_ctx *ctx);
/* tile i -> {x, y, w, h} (pixels); returns non-zero if i is out of range */
int xpnghip_ctx_tile(const xpnghip_ctx *ctx, uint64_t i, uint64_t xywh[4]);
/* upper bound of the concatenated blobs of tiles [t0, t1) (raw fallback bound: sum(w*h*pxsz + 4)) */
uint64_t xpnghip_ctx_blob_bound(const xpnghip_ctx *ctx, uint64_t t0, uint64_t t1);
uint64_t xpnghip_ctx_workspace_bytes(const xpnghip_ctx *ctx);

/* Encode tiles [t0, t1) of the device raster `d_raster` (full w*h*pxsz image, device pointer) into
 * `d_blobs` (device pointer, capacity >= xpnghip_ctx_blob_bound).  On return *blobs_len is the byte
 * count (one stream sync).  Pass blobs_len == NULL to skip the sync and read the length later with
 * xpnghip_ctx_last_blobs_len() after synchronising the stream yourself. */
int xpnghip_encode_device(xpnghip_ctx *ctx, int mode, const void *d_raster, uint64_t t0, uint64_t t1,
                          void *d_blobs, uint64_t *blobs_len, void *stream);
uint64_t xpnghip_ctx_last_blobs_len(xpnghip_ctx *ctx);
/* Batched form: nimg <= batch device rasters in, nimg device blob buffers out (host arrays of device pointers);
 * blobs_len, if not NULL, receives nimg lengths after one stream sync. */
int xpnghip_encode_device_batch(xpnghip_ctx *ctx, int mode, const void *const *d_rasters, uint32_t nimg, uint64_t t0,
                                uint64_t t1, void *const *d_blobs, uint64_t *blobs_len, void *stream);
uint64_t xpnghip_ctx_last_blobs_len_at(xpnghip_ctx *ctx, uint32_t img);

/* Decode tiles [t0, t1).  d_blobs holds their concatenated blobs (device); tile_off[i - t0] is the byte
 * offset of tile i's blob inside d_blobs (host array from the serial size walk, libxpng.c:982), or tile_off == NULL:
 * the walk is done on the device (one lane per image follows the 24-bit sizes), so a caller whose blobs live in HBM never
 * copies them back to find the offsets.
 * Blob buffers handed to these entry points need 64 readable bytes behind their contents (their last words are fetched in
 * aligned blocks); the host-side wrappers (xpnghip_encode_tiles / xpnghip_decode_tiles / libxpng.so) allocate that
 * themselves.  A WHOLE raster needs no slack: every kernel that reads one in 16-byte pieces clamps at w*h*pxsz bytes.  A caller
 * that hands over only a BAND of the raster behind a virtual base pointer (a rank coding tile range [t0, t1) keeps rows
 * [y0, y1) and passes band - y0*w*pxsz) must keep 16 readable bytes behind the band's last row unless the band ends with the
 * raster: the clamp is at the end of the whole raster, so the last 16-byte piece of the band's last row may reach up to 15 bytes
 * into the next row (bench.py and the multi-device wrappers allocate a spare 16 bytes for this). */
int xpnghip_decode_device(xpnghip_ctx *ctx, int mode, const void *d_blobs, uint64_t blobs_len,
                          const uint64_t *tile_off, uint64_t t0, uint64_t t1, void *d_raster, void *stream);

/* Batched form: blobs_len[nimg] = bytes of each blob buffer; tile_off holds nimg * (t1 - t0) offsets, image-major, each
 * relative to its image's blob buffer. */
int xpnghip_decode_device_batch(xpnghip_ctx *ctx, int mode, const void *const *d_blobs, const uint64_t *blobs_len, uint32_t nimg,
                                const uint64_t *tile_off, uint64_t t0, uint64_t t1, void *const *d_rasters, void *stream);
/* The reference decoder trusts the file; this one validates every tile header (lengths, offsets, symbol counts) on the
 * device before using it.  Synchronises `stream` and returns 0 if the last decode accepted every tile, 1 if some tile
 * was rejected (its pixels are left untouched), -1 on a HIP error.  xpnghip_decode_tiles checks it for you. */
int xpnghip_ctx_decode_status(xpnghip_ctx *ctx, void *stream);

/* Stage-only run for BASELINE config 2: predictor chooser + per-pixel transform (libxpng.c:92-140 and
 * the arithmetic of 497-519) over tiles [t0, t1); symbol planes stay in the context's workspace. */
int xpnghip_m1_transform_device(xpnghip_ctx *ctx, const void *d_raster, uint64_t t0, uint64_t t1, void *stream);
int xpnghip_m1_transform_device_batch(xpnghip_ctx *ctx, const void *const *d_rasters, uint32_t nimg, uint64_t t0,
                                      uint64_t t1, void *stream);

/* ---- introspection for parity tests (copies intermediates of the LAST encode to host) --------------
 * what: 0 = predictor byte pr (1 B), 1..5 = planes nl,r,g,b,a (w*h B each, tile-linear),
 *       10..18 = context stream 0..8, 19 = residual bit-stream words k, 20..29 = rANS block 0..9,
 *       30 = chooser cost sums (16 B).  Returns bytes written to `out` (<= cap) or -1. */
int64_t xpnghip_debug_fetch(xpnghip_ctx *ctx, int what, uint64_t tile, void *out, uint64_t cap);

/* ---- wave probe for placement studies (tools/wave_probe.py; no reference counterpart) ---------------
 * Exists only in libxpng_hip_probes.so (xpnghip_probes_built() == 1); in the release library the two calls fail.
 * Registers a device buffer of `cap` 32-byte records {u32 kernel, block, HW_ID, XCC_ID; u64 t0, t1 (100 MHz)}: wave 0 of every
 * workgroup of the serial-chain kernels appends one when it ends.  d_buf == NULL switches the probe off. */
int xpnghip_debug_probe(void *d_buf, uint32_t cap);
int64_t xpnghip_debug_probe_count(void);
int xpnghip_probes_built(void);

#ifdef __cplusplus
}
#endif
#endif
