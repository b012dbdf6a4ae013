/* Test driver: runs the EXR and PNG decoders of csrc/egoprior_codec.c over every file named on the command line, each
 * file in a heap buffer of exactly its size and each output in a buffer of exactly H*W elements, so that a build with
 * -fsanitize=address,undefined reports any read or write outside them.  Prints one line per file: name, return codes. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int epc_exr_half_z_size(const uint8_t* buf, size_t n, int32_t* H, int32_t* W);
int epc_exr_half_z_decode(const uint8_t* buf, size_t n, uint16_t* out, int32_t H, int32_t W);
int epc_png_gray8_size(const uint8_t* buf, size_t n, int32_t* H, int32_t* W);
int epc_png_gray8_decode(const uint8_t* buf, size_t n, uint8_t* out, int32_t H, int32_t W);

int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t* buf = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
        if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 2; }
        fclose(f);
        int32_t H = 0, W = 0;
        int rc_size, rc_dec = -100;
        const size_t len = strlen(argv[a]);
        const int is_png = len > 4 && strcmp(argv[a] + len - 4, ".png") == 0;
        if (is_png) {
            rc_size = epc_png_gray8_size(buf, (size_t)n, &H, &W);
            if (rc_size == 0) {
                uint8_t* out = (uint8_t*)malloc((size_t)H * (size_t)W);
                rc_dec = epc_png_gray8_decode(buf, (size_t)n, out, H, W);
                free(out);
            }
        } else {
            rc_size = epc_exr_half_z_size(buf, (size_t)n, &H, &W);
            if (rc_size == 0) {
                uint16_t* out = (uint16_t*)malloc(2 * (size_t)H * (size_t)W);
                rc_dec = epc_exr_half_z_decode(buf, (size_t)n, out, H, W);
                free(out);
            }
        }
        printf("%s %d %d\n", argv[a], rc_size, rc_dec);
        free(buf);
    }
    return 0;
}
