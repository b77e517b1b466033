/*
 * oracle/convert_oracle.c -- CPU restatement of GStreamer's `videoconvert` for I420 <-> RGBA, the
 * element the reference's own colorlut example wraps around the filter
 * (`... ! videoconvert ! colorlut location=... ! videoconvert ! ...`, video/colorlut/src/colorlut/imp.rs:17-19;
 * SURVEY.md 8f-3).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * videoconvert lives in gst-plugins-base (gst-libs/gst/video/video-converter.c, video-chroma.c,
 * video-orc.orc), which is NOT under /root/reference.  This restates what its 1.14.0 build does for
 * these two conversions with default caps (no colorimetry / chroma-site fields), and it is PINNED against
 * that build itself: tests/golden/make_videoconvert_golden.py runs the image's gst-launch-1.0
 * (GStreamer 1.14.0) on seeded frames and tests/test_convert_oracle_cpu.py compares byte for byte.
 *
 *  I420 -> RGBA: the `convert_I420_pack_ARGB` fast path = orc program video_orc_convert_I420_BGRA/ARGB:
 *     chroma upsampled by duplication (loadupdb; row y uses chroma row y/2), then per pixel in
 *     saturating 16-bit arithmetic
 *        wy = mulhsw(splatbw(Y-128), p1);  R = convssswb(addssw(wy, mulhsw(splatbw(V-128), p2)))
 *        B = convssswb(addssw(wy, mulhsw(splatbw(U-128), p3)))
 *        G = convssswb(addssw(addssw(wy, mulhsw(splatbw(U-128), p4)), mulhsw(splatbw(V-128), p5)))
 *     and +128 on every byte; alpha 255.  p1..p5 = rint(256 * matrix): BT.601 {298,409,516,-100,-208}
 *     for frames of <= 576 lines, BT.709 {298,459,541,-55,-136} up to 2159 lines, BT.2020 {298,430,548,-48,-167}
 *     from 2160 lines (gst_video_info_set_format's default colorimetry in 1.14.0).
 *  RGBA -> I420: generic path: video_orc_matrix8 per pixel, c = clamp(((a*R + b*G + c*B) >> 8) + offset),
 *     BT.601 Y{66,129,25}+16 U{-38,-74,112}+128 V{112,-94,-18}+128, BT.709 Y{47,157,16} U{-26,-87,112}
 *     V{112,-102,-10}, BT.2020 Y{58,149,13} U{-31,-81,112} V{112,-103,-9}; then chroma down-sampling, VERTICAL first ((a + b + 1) >> 1 of rows 2j, 2j+1), then
 *     horizontal: chroma-site none (<= 576 lines) (a + b + 1) >> 1; h-cosited (> 576 lines)
 *     video_chroma_down_h2_cs_u8: first (3a + b + 2) >> 2, interior (l + 2c + r + 2) >> 2 at even x, last
 *     (l + 3c + 2) >> 2.  Odd width / height: the last column / row is replicated to the next even size first (probed
 *     against the element at 65x33, 3x3, 5x7, 7x601, 66x33, 65x34: byte-identical), the chroma planes have
 *     RU2(w)/2 x RU2(h)/2 samples.
 *  RGBA -> NV12: the same Y, U, V (the element's generic path packs them as Y plane + interleaved UV plane; probed at
 *     64x32 ... 1280x720 incl. odd sizes and all three colorimetry defaults).
 *  NV12 -> RGBA is NOT I420 -> RGBA with de-interleaved chroma: 1.14.0 has no NV12 fast path; its generic path unpacks to
 *     AYUV (chroma duplicated), up-samples the chroma HORIZONTALLY first (video-chroma.c: chroma-site none, <= 576 lines:
 *     pixel pairs (i, i+1), i odd, i < w-1, become (3a + b + 2) >> 2 and (a + 3b + 2) >> 2 of the two chroma samples around
 *     them, pixel 0 and an odd last pixel keep their sample; h-cosited, > 576 lines: odd pixels i < w-1 become (a + b + 1) >> 1
 *     of their neighbours), then VERTICALLY (row pairs (j, j+1), j odd, j < h-1: (3a + b + 2) >> 2 and (a + 3b + 2) >> 2 of the
 *     horizontally interpolated chroma rows around them; row 0 and an odd last row keep theirs), then applies the same
 *     saturating 16-bit matrix as the I420 fast path.  Derived by probing the element (order of the two passes included) and
 *     pinned against it: 64x32, 66x34, 65x33, 7x5, 2x2, 4x6, 16x578, 8x2160, 1280x720 (tests/golden/videoconvert_kat.npz).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

static int clamp_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* orc: subb x,128 ; splatbw */
static int splat_s16(int byte_value)
{
    int ub = (byte_value - 128) & 0xff;
    int w = (ub << 8) | ub;
    return w >= 32768 ? w - 65536 : w;
}

static int mulhsw(int a, int b) { return (a * b) >> 16; } /* arithmetic shift of the signed 32-bit product */
static int addssw(int a, int b) { return clamp_i(a + b, -32768, 32767); }
static int convssswb(int a) { return clamp_i(a, -128, 127); }

/* 1 BT.601 (chroma-site none), 2 BT.709 (h-cosited), 3 BT.2020 (h-cosited); 0 = gst_video_info_set_format's default
 * for a frame of `height` lines in GStreamer 1.14.0: <= 576 SD, < 2160 HD, else UHD (probed with 8-pixel-wide frames of
 * 576 / 578 / 2158 / 2160 lines and 3840x8: only the height decides) */
static int pick_standard(uint32_t height, int standard)
{
    if (standard >= 1 && standard <= 3) return standard;
    return height <= 576 ? 1 : (height < 2160 ? 2 : 3);
}

int orc_convert_i420_to_rgba(const uint8_t *y_plane, const uint8_t *u_plane, const uint8_t *v_plane, uint32_t y_stride,
                             uint32_t u_stride, uint32_t v_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *rgba, uint32_t rgba_stride)
{
    static const int k[3][5] = {{298, 409, 516, -100, -208}, {298, 459, 541, -55, -136}, {298, 430, 548, -48, -167}};
    const int *p = k[pick_standard(height, standard) - 1];
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            int wy = mulhsw(splat_s16(y_plane[(size_t)y * y_stride + x]), p[0]);
            int wu = splat_s16(u_plane[(size_t)(y / 2) * u_stride + x / 2]);
            int wv = splat_s16(v_plane[(size_t)(y / 2) * v_stride + x / 2]);
            int r = convssswb(addssw(wy, mulhsw(wv, p[1])));
            int b = convssswb(addssw(wy, mulhsw(wu, p[2])));
            int g = convssswb(addssw(addssw(wy, mulhsw(wu, p[3])), mulhsw(wv, p[4])));
            uint8_t *o = rgba + (size_t)y * rgba_stride + (size_t)x * 4;
            o[0] = (uint8_t)(r + 128); o[1] = (uint8_t)(g + 128); o[2] = (uint8_t)(b + 128); o[3] = 255;
        }
    return ORC_OK;
}

int orc_convert_rgba_to_i420(const uint8_t *rgba, uint32_t rgba_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *y_plane, uint8_t *u_plane, uint8_t *v_plane, uint32_t y_stride, uint32_t u_stride,
                             uint32_t v_stride)
{
    static const int mats[3][3][3] = {{{66, 129, 25}, {-38, -74, 112}, {112, -94, -18}},
                                      {{47, 157, 16}, {-26, -87, 112}, {112, -102, -10}},
                                      {{58, 149, 13}, {-31, -81, 112}, {112, -103, -9}}};
    if (width == 0 || height == 0)
        return ORC_ERR_PANIC;
    const int std_ = pick_standard(height, standard);
    const int hd = std_ != 1; /* HD and UHD defaults carry chroma-site h-cosited */
    const int (*m)[3] = mats[std_ - 1];
    const uint32_t we = (width + 1) & ~1u, he = (height + 1) & ~1u; /* the frame with its last column / row replicated */
#define PX(yy, xx) (rgba + (size_t)((yy) < height ? (yy) : height - 1) * rgba_stride + (size_t)((xx) < width ? (xx) : width - 1) * 4)
    int *v2 = (int *)malloc(sizeof(int) * we); /* one vertically averaged chroma row */
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const uint8_t *q = PX(y, x);
            y_plane[(size_t)y * y_stride + x] = (uint8_t)clamp_i(((m[0][0] * q[0] + m[0][1] * q[1] + m[0][2] * q[2]) >> 8) + 16, 0, 255);
        }
    for (int c = 1; c <= 2; c++) {
        uint8_t *plane = c == 1 ? u_plane : v_plane;
        uint32_t stride = c == 1 ? u_stride : v_stride;
        for (uint32_t j = 0; j < he / 2; j++) {
            for (uint32_t x = 0; x < we; x++) {
                const uint8_t *q0 = PX(2 * j, x), *q1 = PX(2 * j + 1, x);
                int a = clamp_i(((m[c][0] * q0[0] + m[c][1] * q0[1] + m[c][2] * q0[2]) >> 8) + 128, 0, 255);
                int b = clamp_i(((m[c][0] * q1[0] + m[c][1] * q1[1] + m[c][2] * q1[2]) >> 8) + 128, 0, 255);
                v2[x] = (a + b + 1) >> 1;
            }
            uint8_t *o = plane + (size_t)j * stride;
            const uint32_t cw = we / 2;
            if (!hd) {
                for (uint32_t i = 0; i < cw; i++) o[i] = (uint8_t)((v2[2 * i] + v2[2 * i + 1] + 1) >> 1);
            } else {
                o[0] = (uint8_t)((3 * v2[0] + v2[1] + 2) >> 2);
                for (uint32_t i = 1; i + 1 < cw; i++) o[i] = (uint8_t)((v2[2 * i - 1] + 2 * v2[2 * i] + v2[2 * i + 1] + 2) >> 2);
                if (cw > 1) o[cw - 1] = (uint8_t)((v2[we - 3] + 3 * v2[we - 2] + 2) >> 2);
            }
        }
    }
#undef PX
    free(v2);
    return ORC_OK;
}

/* RGBA -> NV12: Y plane + one plane of interleaved (U, V) pairs, RU2(w)/2 pairs x RU2(h)/2 rows */
int orc_convert_rgba_to_nv12(const uint8_t *rgba, uint32_t rgba_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *y_plane, uint8_t *uv_plane, uint32_t y_stride, uint32_t uv_stride)
{
    const uint32_t cw = (width + 1) / 2, ch = (height + 1) / 2;
    uint8_t *u = (uint8_t *)malloc((size_t)cw * ch), *v = (uint8_t *)malloc((size_t)cw * ch);
    int rc = orc_convert_rgba_to_i420(rgba, rgba_stride, width, height, standard, y_plane, u, v, y_stride, cw, cw);
    if (rc == ORC_OK)
        for (uint32_t j = 0; j < ch; j++)
            for (uint32_t i = 0; i < cw; i++) {
                uv_plane[(size_t)j * uv_stride + 2 * i] = u[(size_t)j * cw + i];
                uv_plane[(size_t)j * uv_stride + 2 * i + 1] = v[(size_t)j * cw + i];
            }
    free(u);
    free(v);
    return rc;
}

/* horizontally up-sampled chroma of row `c` (cw samples, interleaved with pitch `step`) at pixel x of w */
static int nv12_chroma_h(const uint8_t *c, uint32_t step, uint32_t x, uint32_t w, int cosited)
{
    const int cur = c[(size_t)(x / 2) * step];
    if (cosited) {
        if ((x & 1) && x < w - 1) return (cur + c[(size_t)(x / 2 + 1) * step] + 1) >> 1;
        return cur;
    }
    if ((x & 1) && x < w - 1) return (3 * cur + c[(size_t)(x / 2 + 1) * step] + 2) >> 2;
    if (!(x & 1) && x >= 2) return (c[(size_t)(x / 2 - 1) * step] + 3 * cur + 2) >> 2;
    return cur;
}

int orc_convert_nv12_to_rgba(const uint8_t *y_plane, const uint8_t *uv_plane, uint32_t y_stride, uint32_t uv_stride, uint32_t width,
                             uint32_t height, int standard, uint8_t *rgba, uint32_t rgba_stride)
{
    static const int k[3][5] = {{298, 409, 516, -100, -208}, {298, 459, 541, -55, -136}, {298, 430, 548, -48, -167}};
    const int std_ = pick_standard(height, standard);
    const int *p = k[std_ - 1];
    const int cosited = std_ != 1;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            int uv[2];
            for (int c = 0; c < 2; c++) {
                const uint8_t *row = uv_plane + (size_t)(y / 2) * uv_stride + c;
                const int cur = nv12_chroma_h(row, 2, x, width, cosited);
                if ((y & 1) && y < height - 1) uv[c] = (3 * cur + nv12_chroma_h(row + uv_stride, 2, x, width, cosited) + 2) >> 2;
                else if (!(y & 1) && y >= 2) uv[c] = (nv12_chroma_h(row - uv_stride, 2, x, width, cosited) + 3 * cur + 2) >> 2;
                else uv[c] = cur;
            }
            int wy = mulhsw(splat_s16(y_plane[(size_t)y * y_stride + x]), p[0]);
            int wu = splat_s16(uv[0]), wv = splat_s16(uv[1]);
            int r = convssswb(addssw(wy, mulhsw(wv, p[1])));
            int b = convssswb(addssw(wy, mulhsw(wu, p[2])));
            int g = convssswb(addssw(addssw(wy, mulhsw(wu, p[3])), mulhsw(wv, p[4])));
            uint8_t *o = rgba + (size_t)y * rgba_stride + (size_t)x * 4;
            o[0] = (uint8_t)(r + 128); o[1] = (uint8_t)(g + 128); o[2] = (uint8_t)(b + 128); o[3] = 255;
        }
    return ORC_OK;
}
