/* xpng.h -- public API of the MI355X-native xPNG library (drop-in for the reference's xpng.h).
 *
 * Same six exported symbols, same argument meaning, same 0 = success / 1 = failure convention as the
 * reference header (reference xpng.h:5-20).  A program written against the reference's xpng.h links
 * against libxpng.so from this repo unchanged; the tile codec underneath runs on the GPU through the
 * C-ABI declared in xpng_hip.h instead of on pthreads.
 */
#ifndef XPNG_H
#define XPNG_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference xpng.h:5-8 */
#define XPNG_COMPRESSION_TYPE_FAST 1
#define XPNG_COMPRESSION_TYPE_SLOW 2
#define XPNG_COMPRESSION_TYPE_EXJPEG 3
#define XPNG_COMPRESSION_TYPE_UNCOMPRESSED 7

/* reference xpng.h:10 -- interleaved 8-bit RGB (A=0) or RGBA (A=1), row-major, s = w*h*(3+A) */
typedef struct xpng_t {
    uint8_t *p;
    uint64_t w, h, s;
    _Bool A;
} xpng_t;

#define XPNG_CHECK __attribute__((warn_unused_result))

/* reference xpng.h:12-13 (libxpng.c:791, 999) */
XPNG_CHECK _Bool xpng_store(uint64_t mode, const xpng_t *pm, const char *xpng);
XPNG_CHECK _Bool xpng_load(const char *xpng, xpng_t *pm);
/* reference xpng.h:15 (libxpng.c:1011): prints "Not Implemented." and fails, as the reference does */
XPNG_CHECK _Bool xpng_from_jpg(const char *jpg, const char *xpng);
/* reference xpng.h:17-20 (libxpng.c:723, 963, 1004).  T = worker count (libxpng.c:146-151); here the workers are the GPUs
 * of this process: T >= 1 uses min(T, visible GPUs, tiles) devices, each coding one contiguous tile range, the blob ranges
 * gathered on the first device for the concatenation; T == 0 (what xpng_store / xpng_load pass, the reference's "nproc")
 * is ONE device (XPNG_GPUS=n in the environment: n): the multi-device form is opt-in until a byte-parity run on real peer
 * GPUs exists (include/xpng_hip.h says the same).  The output bytes do not depend on T. */
XPNG_CHECK _Bool xpng_store_T(uint64_t T, uint64_t mode, const xpng_t *pm, const char *xpng);
XPNG_CHECK _Bool xpng_load_T(uint64_t T, const char *xpng, xpng_t *pm);
XPNG_CHECK _Bool xpng_from_jpg_T(uint64_t T, const char *jpg, const char *xpng);

/* reference 7/seven.h:3-4 (7/libseven.c:3-36): the `.7` raw container */
_Bool store_7(const xpng_t *pm, const char *fn);
_Bool load_7(const char *fn, xpng_t *pm);

#ifdef __cplusplus
}
#endif
#endif
