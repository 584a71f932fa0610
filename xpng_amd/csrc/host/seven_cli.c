/* seven_cli.c -- the `seven` command: PNG <-> `.7` (reference 7/seven.c:39-79, same arguments, usage text and exit status).
 *
 * The reference links libpng through <png.h>.  This image ships the libpng16 runtime but no headers, so the four
 * "simplified API" entry points are bound at run time (dlopen) against the documented public ABI of libpng 1.6:
 * png_image, PNG_IMAGE_VERSION, PNG_FORMAT_*.  If libpng16 is not present the command fails with a message; nothing
 * else in the library depends on it.
 *
 *   --to_7   : read the PNG as 8-bit RGB or RGBA (alpha iff the file has it), then normalize_RGBA exactly as xpng_store
 *              does (7/seven.c:4-37 duplicates libxpng.c:688-721) and write the `.7`
 *   --to_png : read the `.7` and write it as an 8-bit RGB / RGBA PNG
 */
#include "../../../include/xpng.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {           /* libpng 1.6 png_image (png.h, "simplified API") */
    void *opaque;
    uint32_t version, width, height, format, flags, colormap_entries, warning_or_error;
    char message[64];
} png_image_abi;
#define PNG_IMAGE_VERSION_ABI 1u
#define PNG_FORMAT_FLAG_ALPHA_ABI 0x01u
#define PNG_FORMAT_FLAG_COLOR_ABI 0x02u
#define PNG_FORMAT_FLAG_LINEAR_ABI 0x04u
#define PNG_FORMAT_RGB_ABI PNG_FORMAT_FLAG_COLOR_ABI
#define PNG_FORMAT_RGBA_ABI (PNG_FORMAT_FLAG_COLOR_ABI | PNG_FORMAT_FLAG_ALPHA_ABI)

typedef int (*begin_read_fn)(png_image_abi *, const char *);
typedef int (*finish_read_fn)(png_image_abi *, const void *background, void *buffer, int32_t row_stride, void *colormap);
typedef int (*write_file_fn)(png_image_abi *, const char *, int convert_to_8bit, const void *buffer, int32_t row_stride, const void *colormap);
typedef void (*free_fn)(png_image_abi *);

static void *png_lib(void) {
    static const char *names[] = {"libpng16.so.16", "libpng16.so", "libpng.so", NULL};
    for (int i = 0; names[i]; i++) { void *h = dlopen(names[i], RTLD_NOW); if (h) return h; }
    fprintf(stderr, "seven: libpng16 runtime not found\n");
    return NULL;
}

/* libxpng.c:688-721 / 7/seven.c:4-37.  Replaces pm->p (malloc'ed) when the raster changes. */
static int normalize_rgba(xpng_t *pm) {
    if (!pm->A) return 0;
    const uint64_t n = pm->w * pm->h;
    const uint8_t *p = pm->p;
    int hidden = 0, translucent = 0;
    for (uint64_t i = 0; i < n; i++, p += 4) {
        if (p[3] == 0 && (p[0] | p[1] | p[2])) { hidden = 1; break; }
        if (p[3] != 255) translucent = 1;
    }
    if (hidden) {
        for (uint64_t i = 0; i < n; i++) if (pm->p[4 * i + 3] == 0) memset(pm->p + 4 * i, 0, 4);
        return 0;
    }
    if (translucent) return 0;
    uint8_t *q = malloc(n * 3);
    if (!q) return 1;
    for (uint64_t i = 0; i < n; i++) { q[3 * i] = pm->p[4 * i]; q[3 * i + 1] = pm->p[4 * i + 1]; q[3 * i + 2] = pm->p[4 * i + 2]; }
    free(pm->p);
    pm->p = q; pm->s = n * 3; pm->A = 0;
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 4 && (!strcmp(argv[1], "--to_7") || !strcmp(argv[1], "--to_png"))) {
        void *lib = png_lib();
        if (!lib) return 1;
        begin_read_fn begin_read = (begin_read_fn)dlsym(lib, "png_image_begin_read_from_file");
        finish_read_fn finish_read = (finish_read_fn)dlsym(lib, "png_image_finish_read");
        write_file_fn write_file = (write_file_fn)dlsym(lib, "png_image_write_to_file");
        free_fn image_free = (free_fn)dlsym(lib, "png_image_free");
        if (!begin_read || !finish_read || !write_file || !image_free) { fprintf(stderr, "seven: libpng16 lacks the simplified API\n"); return 1; }
        png_image_abi img;
        memset(&img, 0, sizeof img);
        img.version = PNG_IMAGE_VERSION_ABI;
        xpng_t pm;
        if (!strcmp(argv[1], "--to_7")) {
            begin_read(&img, argv[2]);
            if (img.warning_or_error > 1) return 1;
            if (img.format & PNG_FORMAT_FLAG_LINEAR_ABI) { image_free(&img); return 1; }  /* 16-bit files are refused, as in the reference */
            img.format = (img.format & PNG_FORMAT_FLAG_ALPHA_ABI) ? PNG_FORMAT_RGBA_ABI : PNG_FORMAT_RGB_ABI;
            pm.w = img.width; pm.h = img.height; pm.A = img.format == PNG_FORMAT_RGBA_ABI;
            pm.s = pm.w * pm.h * (3u + pm.A);
            pm.p = malloc(pm.s ? pm.s : 1);
            if (!pm.p) { image_free(&img); return 1; }
            finish_read(&img, NULL, pm.p, 0, NULL);
            if (img.warning_or_error > 1) return 1;
            return (int)(normalize_rgba(&pm) || store_7(&pm, argv[3]));
        }
        if (load_7(argv[2], &pm)) return 1;
        img.width = (uint32_t)pm.w; img.height = (uint32_t)pm.h;
        img.format = pm.A ? PNG_FORMAT_RGBA_ABI : PNG_FORMAT_RGB_ABI;
        write_file(&img, argv[3], 0, pm.p, 0, NULL);
        return (int)(img.warning_or_error > 1);
    }
    printf("\n"
           "./seven --to_7   example.png example.7\n"
           "./seven --to_png example.7   example.png\n"
           "\n");
    return 1;
}
