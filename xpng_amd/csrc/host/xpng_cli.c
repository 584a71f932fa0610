/* xpng_cli.c -- the `xpng` command (reference xpng.c:3-24): same arguments, usage text and exit status. */
#include "../../../include/xpng.h"

#include <stdio.h>
#include <string.h>

int main(int argc, char **argv) {
    xpng_t pm;
    if (argc == 4 && argv[1][0] == '-' && strlen(argv[1]) == 2) {
        switch (argv[1][1]) {
        case '1': case '2': case '7':
            return (int)(load_7(argv[2], &pm) || xpng_store((uint64_t)(argv[1][1] - '0'), &pm, argv[3]));
        case '3':
            return (int)xpng_from_jpg(argv[2], argv[3]);
        case 'd':
            return (int)(xpng_load(argv[2], &pm) || store_7(&pm, argv[3]));
        default: break;
        }
    }
    printf("\n"
           "encode: ./xpng -[127] example.7    example.xpng\n"
           "        ./xpng -3     example.jpg  example.xpng\n"
           "decode: ./xpng -d     example.xpng example.7\n"
           "\n");
    return 1;
}
