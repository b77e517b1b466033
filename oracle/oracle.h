/*
 * oracle/oracle.h -- CPU restatement of the reference's per-pixel loops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gst-plugin-rs_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / reported baseline.
 *
 * The reference (sdroege/gst-plugin-rs) is Rust and cannot be built in this image
 * (no rustc/cargo), so this is a "port" oracle: plain C that follows the reference
 * loops statement by statement.  Each function cites the file:line it restates.
 * Built with `gcc -O3 -ffp-contract=off` (no fast-math) so that every f32 operation
 * rounds exactly once, as rustc emits it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   hsv      -- pinned by the reference's own unit vectors, video/hsv/src/hsvutils.rs:219-279
 *   colorlut -- parser pinned by video/colorlut/src/parser.rs:377-474; the reference has no
 *               pixel-output test for the apply_ / sample_ functions => pixel arithmetic "parity unpinned"
 *               beyond a second independent numpy-f32 restatement (tests/np_twin.py)
 *   colordetect / videocompare / roundedcorners -- arithmetic lives in third-party crates that
 *               are not under /root/reference (color-thief 0.2.2, color-name 1.2.0,
 *               image_hasher 3.1.1, cairo).  Pinned only by the reference pipeline tests
 *               (solid red => "red"; identical frames => distance 0).  Otherwise "parity unpinned".
 */
#ifndef MI355VFX_ORACLE_H
#define MI355VFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric values as mvfx_format in include/mi355vfx.h (kept in sync by
 * tests/test_abi.py); restated here so the oracle has no dependency on the product. */
enum {
    ORC_FORMAT_RGBX = 0,
    ORC_FORMAT_XRGB = 1,
    ORC_FORMAT_BGRX = 2,
    ORC_FORMAT_XBGR = 3,
    ORC_FORMAT_RGBA = 4,
    ORC_FORMAT_ARGB = 5,
    ORC_FORMAT_BGRA = 6,
    ORC_FORMAT_ABGR = 7,
    ORC_FORMAT_RGB = 8,
    ORC_FORMAT_BGR = 9,
    ORC_FORMAT_RGBA64_LE = 10,
    ORC_FORMAT_RGBA64_BE = 11,
    ORC_FORMAT_I420 = 12,
    ORC_FORMAT_A420 = 13,
    ORC_FORMAT_RGB10A2_LE = 14, /* colorlut only: the third format of d3d12colorlut's caps (d3d12colorlut/imp.rs:236-244) */
    ORC_FORMAT_NV12 = 15        /* converter output only */
};

/* Error codes of the oracle (negative).  ORC_ERR_PANIC marks inputs on which the
 * reference would panic (assert_eq!/unreachable!). */
#define ORC_OK 0
#define ORC_ERR_PANIC (-1)
#define ORC_ERR_FORMAT (-2)
#define ORC_ERR_PARSE (-3)

/* ---- video/hsv/src/hsvutils.rs ---- */
void orc_hsv_from_rgb(const uint8_t in_p[3], float out[3]); /* :44-84 */
void orc_hsv_from_bgr(const uint8_t in_p[3], float out[3]); /* :88-128 */
void orc_hsv_to_rgb(const float in_p[3], uint8_t out[3]);   /* :132-163 */
void orc_hsv_to_bgr(const float in_p[3], uint8_t out[3]);   /* :167-198 */

/* ---- video/hsv/src/hsvfilter/imp.rs:76-120 + :322-377 ----
 * settings = {hue_shift, saturation_mul, saturation_off, value_mul, value_off} */
int orc_hsvfilter_transform_frame_ip(uint8_t *data, size_t data_len, uint32_t width,
                                     uint32_t stride, int format, const float settings[5]);

/* ---- video/hsv/src/hsvdetector/imp.rs:100-160 + :422-707 ----
 * settings = {hue_ref, hue_var, saturation_ref, saturation_var, value_ref, value_var} */
int orc_hsvdetector_transform_frame(const uint8_t *in_data, size_t in_len, uint32_t in_stride,
                                    int in_format, uint8_t *out_data, size_t out_len,
                                    uint32_t out_stride, int out_format, uint32_t width,
                                    const float settings[6]);

/* Debug helper used by the f32 parity test: from_rgb for every pixel of a packed
 * RGBx frame, writing 3 floats per pixel. */
void orc_hsv_from_rgb_frame(const uint8_t *rgbx, size_t n_pixels, float *hsv_out);

/* ---- video/colorlut/src/parser.rs ---- */
typedef struct orc_cube_lut orc_cube_lut;
/* returns NULL on error; err (if non-NULL, cap bytes) receives the message */
orc_cube_lut *orc_cube_parse(const char *text, size_t len, char *err, size_t cap);
void orc_cube_free(orc_cube_lut *lut);
int orc_cube_is_3d(const orc_cube_lut *lut);
uint32_t orc_cube_size(const orc_cube_lut *lut);
const float *orc_cube_domain_scale(const orc_cube_lut *lut);
const float *orc_cube_domain_offset(const orc_cube_lut *lut);
/* 3-D: size^3 * 4 floats ([r,g,b,1.0], R fastest); 1-D: NULL */
const float *orc_cube_rgba(const orc_cube_lut *lut);
/* 1-D: channel table c (0..2), size floats; 3-D: NULL */
const float *orc_cube_table_1d(const orc_cube_lut *lut, int c);

/* ---- video/colorlut/src/colorlut/imp.rs:226-543 ---- */
int orc_colorlut_transform_frame(const orc_cube_lut *lut, const uint8_t *src, size_t src_len,
                                 uint32_t src_stride, uint8_t *dst, size_t dst_len,
                                 uint32_t dst_stride, uint32_t width, uint32_t height,
                                 int format);

/* ---- video/videofx: third-party-backed algorithms (restated from the crates'
 * published behaviour; "self-golden") ---- */

/* color-thief 0.2.2 get_palette(): call site colordetect/imp.rs:68-74.
 * pixels = whole plane incl. padding (flat), format in {RGB,RGBA,ARGB,BGR,BGRA}.
 * palette_out receives up to max_colors packed 0x00RRGGBB; returns count or <0. */
int orc_colordetect_histogram(const uint8_t *pixels, size_t len, int format, uint32_t quality,
                              int32_t *hist32768, uint32_t minmax[6], uint64_t *n_counted);
int orc_colordetect_palette(const uint8_t *pixels, size_t len, int format, uint32_t quality,
                            uint32_t max_colors, uint32_t *palette_out);
int orc_mmcq_from_histogram(const int32_t *hist32768, const uint32_t minmax[6],
                            uint32_t max_colors, uint32_t *palette_out);
/* color-name 1.2.0 css::Color::similar(): call site colordetect/imp.rs:77-79; returns a
 * static lower-cased name */
const char *orc_css_color_similar(uint8_t r, uint8_t g, uint8_t b);

/* image_hasher 3.1.1 Blockhash (8x8 bits): call sites videocompare/hashed_image.rs:37-45,70 */
int orc_blockhash_sums(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride,
                       int format, uint32_t sums[64]);
uint64_t orc_blockhash_bits(const uint32_t sums[64], uint32_t width, uint32_t height);
int orc_blockhash(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride,
                  int format, uint64_t *hash);
uint32_t orc_hamming64(uint64_t a, uint64_t b);

/* imagersoverlay's per-frame work: `composition.blend(frame)` (video/image/src/overlay/imp.rs:703-727) ==
 * gst_video_overlay_composition_blend -> gst_video_blend of libgstvideo (NOT under /root/reference).  One unscaled
 * BGRA rectangle (non-premultiplied, what load_image builds: imp.rs:241-283) at (x, y) with the rectangle's global alpha
 * onto a packed RGB frame.  Restated from and pinned against the image's own libgstvideo 1.14.0 (tests/golden/
 * make_overlay_blend_golden.py), like the videoconvert restatement. */
int orc_overlay_blend(uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format,
                      const uint8_t *overlay_bgra, uint32_t overlay_width, uint32_t overlay_height, uint32_t overlay_stride,
                      int32_t x, int32_t y, float global_alpha);
/* image_hasher 3.1.1 Mean / Gradient / VertGradient / DoubleGradient on image 0.25.10's grayscale +
 * Lanczos3 resize (hashed_image.rs:89-107); algo = GstVideoCompareHashAlgorithm value 0..3.
 * PARITY UNPINNED (crates not under /root/reference). */
int orc_gray_resize_lanczos3(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format,
                             uint32_t nw, uint32_t nh, uint8_t *out);
int orc_image_hash(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format, int algo,
                   uint64_t *hash, uint32_t *n_bits);

/* GStreamer 1.14.0 videoconvert I420 <-> RGBA with default caps (convert_oracle.c; gst-plugins-base is not under
 * /root/reference; PINNED by goldens made with the image's own GStreamer 1.14.0).  standard: 0 by height,
 * 1 BT.601 + chroma-site none, 2 BT.709 + h-cosited, 3 BT.2020 + h-cosited. */
int orc_convert_i420_to_rgba(const uint8_t *y_plane, const uint8_t *u_plane, const uint8_t *v_plane, uint32_t y_stride,
                             uint32_t u_stride, uint32_t v_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *rgba, uint32_t rgba_stride);
int orc_convert_rgba_to_i420(const uint8_t *rgba, uint32_t rgba_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *y_plane, uint8_t *u_plane, uint8_t *v_plane, uint32_t y_stride, uint32_t u_stride,
                             uint32_t v_stride);
int orc_convert_rgba_to_nv12(const uint8_t *rgba, uint32_t rgba_stride, uint32_t width, uint32_t height, int standard,
                             uint8_t *y_plane, uint8_t *uv_plane, uint32_t y_stride, uint32_t uv_stride);
int orc_convert_nv12_to_rgba(const uint8_t *y_plane, const uint8_t *uv_plane, uint32_t y_stride, uint32_t uv_stride, uint32_t width,
                             uint32_t height, int standard, uint8_t *rgba, uint32_t rgba_stride);

/* SSIM-family distance behind hash-algo=dssim (dssim-core 3.4.0, non-default feature; PARITY
 * UNPINNED, see ssim_oracle.c): f64, formats RGB / RGBA. */
int orc_ssim_distance(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t stride_a,
                      uint32_t stride_b, int format, double *distance, double *per_scale);
/* Row-band partial sums of the same maps (mean == NULL: map sums + counts; else |map - mean| sums). */
int orc_ssim_band(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t stride_a,
                  uint32_t stride_b, int format, uint32_t row_begin, uint32_t row_end, const double *mean,
                  double *sums, double *counts, int *n_scales);
double orc_ssim_combine(const double *mean, const double *mad, int n_scales);

#ifdef __cplusplus
}
#endif
#endif
