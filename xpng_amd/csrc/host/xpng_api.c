/* xpng_api.c -- host driver of the MI355X xPNG library: the drop-in for the reference's
 * xpng_store_T / xpng_load_T (libxpng.c:723-789, 963-997).
 *
 * Everything the reference driver does around its two thread fan-outs stays here, in C, with the same
 * observable behaviour: validation, normalize_RGBA, the `s <= 4 -> level 7` rule, the whole-image
 * single-colour shortcut of level 2, the RGBA level-2 -> level-1 fallback, the file header, the
 * "compressed >= raw -> rewrite as level 7" rule, and the stdout MPx/s line.  The fan-outs themselves
 * (libxpng.c:758, 983) are replaced by xpnghip_encode_tiles / xpnghip_decode_tiles (include/xpng_hip.h),
 * which run on the GPU.  There is no CPU codec in this library: if no HIP device is usable the call
 * fails (returns 1), except for the paths that never reach the tile codec (level 7, single colour).
 */
#include "../../../include/xpng.h"
#include "../../../include/xpng_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define XPNG_MAX_DIM (1u << 24)

static uint64_t now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* libxpng.c:688-721.  Returns 1 on allocation failure.  *out stays NULL when the raster is kept. */
static int normalize_rgba(const xpng_t *in, uint8_t **out, uint64_t *s, _Bool *A) {
    *out = NULL; *s = in->s; *A = in->A;
    if (!in->A) return 0;
    const uint64_t n = in->w * in->h;
    const uint8_t *p = in->p;
    int hidden = 0, translucent = 0;
    for (uint64_t i = 0; i < n; i++, p += 4) {
        if (p[3] == 0 && (p[0] | p[1] | p[2])) { hidden = 1; break; }
        if (p[3] != 255) translucent = 1;
    }
    if (hidden) {
        uint8_t *q = malloc(in->s);
        if (!q) return 1;
        p = in->p;
        for (uint64_t i = 0; i < n; i++, p += 4) {
            if (p[3]) memcpy(q + 4 * i, p, 4); else memset(q + 4 * i, 0, 4);
        }
        *out = q;
        return 0;
    }
    if (translucent) return 0;
    uint8_t *q = malloc(n * 3);
    if (!q) return 1;
    p = in->p;
    for (uint64_t i = 0; i < n; i++, p += 4) { q[3 * i] = p[0]; q[3 * i + 1] = p[1]; q[3 * i + 2] = p[2]; }
    *out = q; *s = n * 3; *A = 0;
    return 0;
}

static int all_pixels_equal(const uint8_t *p, uint64_t n, int pxsz) { /* whole-image form of libxpng.c:628-643; stops at the first difference */
    for (uint64_t i = 1; i < n; i++) if (memcmp(p, p + i * (uint64_t)pxsz, (size_t)pxsz)) return 0;
    return 1;
}

static _Bool write_file(const char *fn, const uint8_t hdr[8], const uint8_t *body, uint64_t len) {
    FILE *f = fopen(fn, "wb");
    if (!f) return 1;
    _Bool bad = fwrite(hdr, 1, 8, f) != 8 || (len && fwrite(body, 1, len, f) != len);
    return (_Bool)(fclose(f) != 0) || bad;
}

static void print_rate(const char *what, uint64_t workers, uint64_t ns, uint64_t pixels) {
    if (!ns) ns = 1; /* same line shape as libxpng.c:761-762 / 986-987 */
    printf("%s, %3d thread%c: %5lu MPx/s\n", what, (int)workers, workers > 1 ? 's' : ' ',
           (unsigned long)((1e9 / (double)ns) * ((double)pixels / 1e6)));
}

/* The tile-codec path of xpng_store_T (levels 1 and 2 on a raster of more than 4 bytes): the raster is uploaded once and
 * normalize_RGBA (libxpng.c:733), the whole-image single-colour test (741-753) and the tile encode (758-769) run on the
 * device.  The normalised raster comes back to the host only for the outputs that contain it verbatim. */
static _Bool store_on_device(uint64_t T, uint64_t mode, const xpng_t *pm, const char *fn, uint64_t t_start) {
    int pxsz = 0;
    xpnghip_image *img = NULL; /* this call's own staging object: concurrent xpng_store calls do not share state */
    if (xpnghip_image_begin(&img, pm->p, pm->w, pm->h, 3 + pm->A, &pxsz)) {
        fprintf(stderr, "xpng: GPU staging failed: %s\n", xpnghip_last_error());
        return 1;
    }
    const _Bool A = pxsz == 4;
    const uint64_t s = pm->w * pm->h * (uint64_t)pxsz;
    _Bool rc = 1;
    uint8_t hdr[8];
    uint8_t *raw = NULL, *blobs = NULL;
    put_u32(hdr, (uint32_t)(pm->w - 1) | ((uint32_t)mode << 24));
    put_u32(hdr + 4, (uint32_t)(pm->h - 1) | ((uint32_t)A << 24));
    if (mode == 2) { /* libxpng.c:741-753 */
        int single = 0;
        if (xpnghip_image_single_colour(img, &single)) goto done;
        if (single) {
            if (!(raw = malloc(s)) || xpnghip_image_fetch(img, raw)) goto done;
            hdr[7] |= 2;
            rc = write_file(fn, hdr, raw, (uint64_t)pxsz);
            goto done;
        }
    }
    if (A && mode == 2) { mode = 1; hdr[3] = 1; } /* libxpng.c:755 */
    if (A && (pm->w < 4 || pm->h < 4)) { /* reference behaviour undefined here: store uncompressed */
        if (!(raw = malloc(s)) || xpnghip_image_fetch(img, raw)) goto done;
        hdr[3] = XPNG_COMPRESSION_TYPE_UNCOMPRESSED;
        rc = write_file(fn, hdr, raw, s);
        goto done;
    }
    {
        uint64_t blen = 0;
        if (xpnghip_image_encode_T(img, T, (int)mode, &blobs, &blen)) {
            fprintf(stderr, "xpng: GPU tile encode failed: %s\n", xpnghip_last_error());
            goto done;
        }
        /* the reference prints its worker-thread count here (libxpng.c:761); the workers of this library are GPUs */
        print_rate("encode", (uint64_t)xpnghip_devices_for(T, pm->w, pm->h), now_ns() - t_start, pm->w * pm->h);
        if (blen >= s) { /* libxpng.c:771-777 */
            if (!(raw = malloc(s)) || xpnghip_image_fetch(img, raw)) goto done;
            hdr[3] = XPNG_COMPRESSION_TYPE_UNCOMPRESSED;
            rc = write_file(fn, hdr, raw, s);
        } else rc = write_file(fn, hdr, blobs, blen);
    }
done:
    xpnghip_image_end(img);
    free(raw);
    free(blobs);
    return rc;
}

_Bool xpng_store_T(uint64_t T, uint64_t mode, const xpng_t *pm, const char *fn) {
    const uint64_t t_start = now_ns();
    /* T: the reference's worker count (libxpng.c:146-151); here the number of GPUs of this process the tile stage may use
     * (0 = automatic), see xpnghip_encode_tiles_T */
    if (!pm || !fn || !pm->p || !pm->w || !pm->h || pm->w > XPNG_MAX_DIM || pm->h > XPNG_MAX_DIM ||
        !(mode == 1 || mode == 2 || mode == 7) || pm->w * pm->h * (3u + pm->A) != pm->s)
        return 1; /* libxpng.c:729-731 */
    /* Everything that reaches the tile codec is staged on the device.  Level 7 and rasters of at most 4 bytes after
     * normalisation (i.e. a single pixel) never do: they are handled here, on the host, and need no GPU. */
    if (mode == 2 && !pm->A && pm->w * pm->h > 1 && all_pixels_equal(pm->p, pm->w * pm->h, 3)) { /* libxpng.c:741-753, RGB input: no GPU needed */
        uint8_t h1[8];
        put_u32(h1, (uint32_t)(pm->w - 1) | (2u << 24));
        put_u32(h1 + 4, (uint32_t)(pm->h - 1));
        h1[7] |= 2;
        return write_file(fn, h1, pm->p, 3);
    }
    if (mode != 7 && pm->w * pm->h > 1) return store_on_device(T, mode, pm, fn, t_start);
    uint8_t *owned = NULL;
    uint64_t s;
    _Bool A;
    if (normalize_rgba(pm, &owned, &s, &A)) return 1;
    const uint8_t *raster = owned ? owned : pm->p;
    uint8_t hdr[8];
    mode = 7; /* explicit level 7, or s <= 4: libxpng.c:735 */
    put_u32(hdr, (uint32_t)(pm->w - 1) | ((uint32_t)mode << 24));
    put_u32(hdr + 4, (uint32_t)(pm->h - 1) | ((uint32_t)A << 24));
    const _Bool rc = write_file(fn, hdr, raster, s);
    free(owned);
    return rc;
}

_Bool xpng_store(uint64_t mode, const xpng_t *pm, const char *fn) { return xpng_store_T(0, mode, pm, fn); }

_Bool xpng_load_T(uint64_t T, const char *fn, xpng_t *pm) {
    if (!fn || !pm) return 1;
    FILE *f = fopen(fn, "rb");
    if (!f) return 1;
    if (fseek(f, 0, SEEK_END)) { fclose(f); return 1; }
    const long fl = ftell(f);
    if (fl < 8 || fseek(f, 0, SEEK_SET)) { fclose(f); return 1; }
    const uint64_t flen = (uint64_t)fl;
    uint8_t *buf = malloc(flen + 16);
    if (!buf) { fclose(f); return 1; }
    if (fread(buf, 1, flen, f) != flen) { fclose(f); free(buf); return 1; }
    fclose(f);
    memset(buf + flen, 0, 16);
    const uint64_t t_start = now_ns(); /* file read is not timed, libxpng.c:967 */
    const uint32_t h0 = get_u32(buf), h1 = get_u32(buf + 4);
    const uint64_t mode = h0 >> 24;
    pm->w = (h0 & 0xFFFFFF) + 1; pm->h = (h1 & 0xFFFFFF) + 1; pm->A = (h1 >> 24) & 1;
    if (!(mode == 1 || mode == 2 || mode == 7)) { free(buf); return 1; }
    const int pxsz = 3 + pm->A;
    pm->s = pm->w * pm->h * (uint64_t)pxsz;
    pm->p = malloc(pm->s);
    if (!pm->p) { free(buf); return 1; }
    _Bool rc = 1;
    if (mode == 7) {
        if (flen >= 8 + pm->s) { memcpy(pm->p, buf + 8, pm->s); rc = 0; }
    } else if (flen == 11u + pm->A && (buf[7] & 2)) { /* libxpng.c:976-980 */
        for (uint64_t i = 0; i < pm->w * pm->h; i++) memcpy(pm->p + i * (uint64_t)pxsz, buf + 8, (size_t)pxsz);
        rc = 0;
    } else {
        if (xpnghip_decode_tiles_T(T, (int)mode, buf + 8, flen - 8, pm->w, pm->h, pxsz, pm->p))
            fprintf(stderr, "xpng: GPU tile decode failed: %s\n", xpnghip_last_error());
        else { print_rate("decode", (uint64_t)xpnghip_devices_for(T, pm->w, pm->h), now_ns() - t_start, pm->w * pm->h); rc = 0; }
    }
    free(buf);
    if (rc) { free(pm->p); pm->p = NULL; }
    return rc;
}

_Bool xpng_load(const char *fn, xpng_t *pm) { return xpng_load_T(0, fn, pm); }

/* libxpng.c:1004-1014: the reference ships this entry point as a stub */
_Bool xpng_from_jpg_T(uint64_t T, const char *jpg, const char *xpng) {
    (void)T; (void)jpg; (void)xpng;
    puts("\nNot Implemented.\n");
    return 1;
}
_Bool xpng_from_jpg(const char *jpg, const char *xpng) { return xpng_from_jpg_T(0, jpg, xpng); }
