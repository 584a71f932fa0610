/* seven.c -- the `.7` raw container (reference 7/libseven.c:3-36): 8-byte header
 * {(w-1) | 7<<24, (h-1) | A<<24} followed by the interleaved raster. */
#include "../../../include/xpng.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

_Bool store_7(const xpng_t *pm, const char *fn) {
    if (!pm || !pm->p || !pm->w || pm->w > (1u << 24) || !pm->h || pm->h > (1u << 24) ||
        pm->s != pm->w * pm->h * (3u + pm->A))
        return 1;
    const uint32_t h[2] = { (uint32_t)(pm->w - 1) | (7u << 24), (uint32_t)(pm->h - 1) | ((uint32_t)pm->A << 24) };
    FILE *f = fopen(fn, "wb");
    if (!f) return 1;
    _Bool bad = fwrite(h, 1, 8, f) != 8 || fwrite(pm->p, 1, pm->s, f) != pm->s;
    return (_Bool)(fclose(f) != 0) || bad;
}

_Bool load_7(const char *fn, xpng_t *pm) {
    struct stat st;
    if (!fn || !pm || stat(fn, &st) || st.st_size < 11) return 1;
    FILE *f = fopen(fn, "rb");
    uint32_t h[2];
    if (!f) return 1;
    if (fread(h, 1, 8, f) != 8) { fclose(f); return 1; }
    pm->w = (h[0] & 0xFFFFFF) + 1; pm->h = (h[1] & 0xFFFFFF) + 1; pm->A = (h[1] >> 24) & 1;
    pm->s = pm->w * pm->h * (3u + pm->A);
    pm->p = NULL;
    if (pm->s + 8 != (uint64_t)st.st_size || (h[0] >> 24) != 7) { fclose(f); return 1; }
    pm->p = malloc(pm->s);
    if (!pm->p || fread(pm->p, 1, pm->s, f) != pm->s) { fclose(f); free(pm->p); pm->p = NULL; return 1; }
    return (_Bool)(fclose(f) != 0);
}
