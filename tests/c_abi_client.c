/*
 * tests/c_abi_client.c -- the boundary used from plain C, the way a cgo / Rust-FFI / C host would:
 * only include/mi355vfx.h and -lmi355vfx.  Built and run by tests/test_c_client.py.
 *
 *   gcc -std=c99 -Wall -Iinclude tests/c_abi_client.c -Lgst-plugin-rs_amd -lmi355vfx -Wl,-rpath,... -o client
 *
 * Exit code 0 = all checks passed; prints "NO_DEVICE" and exits 0 after the host-only checks when
 * no GPU is visible (the compute entry points must then fail loudly, never fall back).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355vfx.h"

#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            fprintf(stderr, "%s:%d: check failed: %s (last error: %s)\n", __FILE__, __LINE__, #cond, mvfx_last_error()); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(void)
{
    CHECK(mvfx_abi_version() == MVFX_ABI_VERSION);

    /* host-only: .cube parsing (video/colorlut/src/parser.rs:381-408) */
    static const char cube[] = "LUT_3D_SIZE 2\n0 0 0\n1 0 0\n0 1 0\n1 1 0\n0 0 1\n1 0 1\n0 1 1\n1 1 1\n";
    mvfx_cube_lut *lut = NULL;
    CHECK(mvfx_cube_lut_parse(cube, sizeof(cube) - 1, &lut) == MVFX_OK);
    CHECK(mvfx_cube_lut_is_3d(lut) && mvfx_cube_lut_size(lut) == 2);
    const float *rgba = mvfx_cube_lut_rgba(lut);
    CHECK(rgba[0] == 0.0f && rgba[3] == 1.0f && rgba[7 * 4 + 0] == 1.0f && rgba[7 * 4 + 2] == 1.0f);
    mvfx_cube_lut *bad = NULL;
    CHECK(mvfx_cube_lut_parse("LUT_1D_SIZE 2\nLUT_3D_SIZE 2\n", 28, &bad) == MVFX_ERR_PARSE && bad == NULL);
    CHECK(strcmp(mvfx_css_color_similar(252, 4, 4), "red") == 0);

    /* 4x2 RGBA frame through hsvfilter with the defaults */
    unsigned char px[32];
    for (int i = 0; i < 32; i++) px[i] = (unsigned char)(i * 37 + 11);
    px[0] = 12; px[1] = 200; px[2] = 77; px[3] = 9; /* SURVEY F5 probe: -> 11,200,76,9 */
    mvfx_frame f = {px, 4, 2, 16, MVFX_FORMAT_RGBA};
    mvfx_hsvfilter_settings s = {0.0f, 1.0f, 0.0f, 1.0f, 0.0f};
    int rc = mvfx_hsvfilter_transform_frame_ip_host(&f, &s);
    if (mvfx_device_count() == 0) {
        CHECK(rc == MVFX_ERR_NO_DEVICE);
        CHECK(strstr(mvfx_last_error(), "no CPU fallback") != NULL);
        CHECK(px[0] == 12 && px[1] == 200 && px[2] == 77); /* untouched */
        mvfx_cube_lut_free(lut);
        printf("NO_DEVICE\n");
        return 0;
    }
    CHECK(rc == MVFX_OK);
    CHECK(px[0] == 11 && px[1] == 200 && px[2] == 76 && px[3] == 9);

    /* identity cube through colorlut: output == input */
    unsigned char in[32], out[32];
    for (int i = 0; i < 32; i++) in[i] = (unsigned char)(255 - i * 7);
    mvfx_frame fi = {in, 4, 2, 16, MVFX_FORMAT_RGBA}, fo = {out, 4, 2, 16, MVFX_FORMAT_RGBA};
    CHECK(mvfx_colorlut_transform_frame_host(lut, &fi, &fo) == MVFX_OK);
    CHECK(memcmp(in, out, 32) == 0);
    CHECK(mvfx_colorlut_transform_frame_host(NULL, &fi, &fo) == MVFX_ERR_NO_LUT);
    mvfx_cube_lut_free(lut);

    /* videocompare: a frame against itself has distance 0 */
    unsigned char big[8 * 8 * 4];
    for (int i = 0; i < (int)sizeof(big); i++) big[i] = (unsigned char)(i * 13);
    mvfx_frame fb = {big, 8, 8, 32, MVFX_FORMAT_RGBA};
    uint64_t h1 = 0, h2 = 1;
    CHECK(mvfx_blockhash_host(&fb, &h1) == MVFX_OK && mvfx_blockhash_host(&fb, &h2) == MVFX_OK);
    CHECK(mvfx_hash_distance(h1, h2) == 0);
    printf("OK\n");
    return 0;
}
