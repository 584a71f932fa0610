/* tool_cli.c -- the `tool` command of the reference's orientation experiments (Mirroring_and_Rotating/tool.c:122-141):
 * one flip / rotation of a `.7` raster, same arguments, usage text and exit status.
 *   --mv  mirror vertically (rows reversed)      --mh  mirror horizontally (columns reversed)     --mvh  both (180 degrees)
 *   --r90 rotate 90 degrees clockwise (pixel (x, y) -> (h-1-y, x); width and height swap)         --r270 = --mvh, then --r90
 *   --tl / --tr are empty in the reference (tool.c:111-117): the raster is written back unchanged.
 * The batched GPU search over the 8 orientations (reference test.rb) is tools/orient_search.py. */
#include "../../../include/xpng.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void px_copy(uint8_t *d, const uint8_t *s, uint64_t z) { memcpy(d, s, z); }

static int remap(xpng_t *pm, int op) { /* 0 r90, 1 r270, 2 mv, 3 mh, 4 mvh */
    const uint64_t z = 3u + pm->A, w = pm->w, h = pm->h;
    uint8_t *out = malloc(pm->s ? pm->s : 1);
    if (!out) return 1;
    for (uint64_t y = 0; y < h; y++)
        for (uint64_t x = 0; x < w; x++) {
            uint64_t nx, ny, nw = w;
            switch (op) {
            case 0: nx = h - 1 - y; ny = x; nw = h; break;          /* clockwise */
            case 1: nx = y; ny = w - 1 - x; nw = h; break;          /* counter-clockwise = mvh then r90 */
            case 2: nx = x; ny = h - 1 - y; break;
            case 3: nx = w - 1 - x; ny = y; break;
            default: nx = w - 1 - x; ny = h - 1 - y; break;
            }
            px_copy(out + (ny * nw + nx) * z, pm->p + (y * w + x) * z, z);
        }
    free(pm->p);
    pm->p = out;
    if (op < 2) { pm->w = h; pm->h = w; }
    return 0;
}

int main(int argc, char **argv) {
    static const char *names[] = {"--r90", "--r270", "--mv", "--mh", "--mvh", "--tl", "--tr"};
    if (argc == 4) {
        int op = -1;
        for (int i = 0; i < 7; i++) if (!strcmp(argv[1], names[i])) op = i;
        if (op >= 0) {
            xpng_t pm;
            if (load_7(argv[2], &pm)) return 1;
            if (op < 5 && remap(&pm, op)) return 1;
            return (int)store_7(&pm, argv[3]);
        }
    }
    printf("\n\t./tool --(r90|r270|mv|mh|mvh|tl|tr) src.7 res.7\n\n");
    return 1;
}
