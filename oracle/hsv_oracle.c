/*
 * oracle/hsv_oracle.c -- CPU restatement of video/hsv (TEST INFRASTRUCTURE ONLY).
 *
 * Follows /root/reference/video/hsv/src/hsvutils.rs, hsvfilter/imp.rs and
 * hsvdetector/imp.rs statement by statement.  Build: gcc -O3 -ffp-contract=off.
 *
 * Rust semantics reproduced here:
 *   f32 `%`            -> fmodf (exact, sign of dividend)
 *   f32::max / min     -> fmaxf / fminf (NaN-ignoring); used by the custom Clamp trait
 *                         hsvutils.rs:16-38  => clamp(NaN, lo, hi) == lo
 *   `as u8` from f32   -> saturating, truncating toward zero, NaN -> 0
 *   no FMA contraction -> -ffp-contract=off
 */
#include "oracle.h"

#include <math.h>
#include <string.h>

static const float EPSILON = 0.00001f; /* hsvutils.rs:40 */

/* hsvutils.rs:16-38 (custom Clamp: self.max(lower).min(upper)) */
static inline float hsv_clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* Rust `f32 as u8` */
static inline uint8_t f32_as_u8(float v)
{
    if (!(v == v))
        return 0;
    if (v <= 0.0f)
        return 0;
    if (v >= 255.0f)
        return 255;
    return (uint8_t)v; /* C conversion truncates toward zero */
}

static inline uint8_t max3_u8(const uint8_t p[3])
{
    uint8_t m = p[0];
    if (p[1] > m) m = p[1];
    if (p[2] > m) m = p[2];
    return m;
}

static inline uint8_t min3_u8(const uint8_t p[3])
{
    uint8_t m = p[0];
    if (p[1] < m) m = p[1];
    if (p[2] < m) m = p[2];
    return m;
}

/* shared body of from_rgb (hsvutils.rs:44-84) and from_bgr (:88-128); they differ only
 * in which byte is called r and which b (:45-47 vs :89-91) */
static inline void from_channels(const uint8_t in_p[3], float r, float g, float b, float out[3])
{
    float value = (float)max3_u8(in_p) / 255.0f;            /* :49-53 */
    float chroma = value - ((float)min3_u8(in_p) / 255.0f); /* :54-59 */

    float hue;
    if (chroma == 0.0f) { /* :61-71 */
        hue = 0.0f;
    } else if (fabsf(value - r) < EPSILON) {
        hue = 60.0f * ((g - b) / chroma);
    } else if (fabsf(value - g) < EPSILON) {
        hue = 60.0f * (2.0f + ((b - r) / chroma));
    } else if (fabsf(value - b) < EPSILON) {
        hue = 60.0f * (4.0f + ((r - g) / chroma));
    } else {
        hue = 0.0f;
    }

    if (hue < 0.0f) /* :73-75 */
        hue += 360.0f;

    float saturation = (value == 0.0f) ? 0.0f : chroma / value; /* :77 */

    out[0] = fmodf(hue, 360.0f);               /* :80 */
    out[1] = hsv_clamp(saturation, 0.0f, 1.0f); /* :81 */
    out[2] = hsv_clamp(value, 0.0f, 1.0f);      /* :82 */
}

void orc_hsv_from_rgb(const uint8_t in_p[3], float out[3])
{
    float r = (float)in_p[0] / 255.0f; /* :45-47 */
    float g = (float)in_p[1] / 255.0f;
    float b = (float)in_p[2] / 255.0f;
    from_channels(in_p, r, g, b, out);
}

void orc_hsv_from_bgr(const uint8_t in_p[3], float out[3])
{
    float b = (float)in_p[0] / 255.0f; /* :89-91 */
    float g = (float)in_p[1] / 255.0f;
    float r = (float)in_p[2] / 255.0f;
    from_channels(in_p, r, g, b, out);
}

/* shared body of to_rgb (:132-163) / to_bgr (:167-198): computes rgb_prime + m */
static inline void to_channels(const float in_p[3], float ch[3])
{
    float c = in_p[2] * in_p[1];        /* :133 */
    float hue_prime = in_p[0] / 60.0f;  /* :134 */
    float x = c * (1.0f - fabsf(fmodf(hue_prime, 2.0f) - 1.0f)); /* :136 */

    float p0, p1, p2; /* :138-154, note the <= comparisons */
    if (hue_prime < 0.0f) {
        p0 = 0.0f; p1 = 0.0f; p2 = 0.0f;
    } else if (hue_prime <= 1.0f) {
        p0 = c; p1 = x; p2 = 0.0f;
    } else if (hue_prime <= 2.0f) {
        p0 = x; p1 = c; p2 = 0.0f;
    } else if (hue_prime <= 3.0f) {
        p0 = 0.0f; p1 = c; p2 = x;
    } else if (hue_prime <= 4.0f) {
        p0 = 0.0f; p1 = x; p2 = c;
    } else if (hue_prime <= 5.0f) {
        p0 = x; p1 = 0.0f; p2 = c;
    } else if (hue_prime <= 6.0f) {
        p0 = c; p1 = 0.0f; p2 = x;
    } else {
        p0 = 0.0f; p1 = 0.0f; p2 = 0.0f;
    }

    float m = in_p[2] - c; /* :156 */
    ch[0] = p0 + m;
    ch[1] = p1 + m;
    ch[2] = p2 + m;
}

void orc_hsv_to_rgb(const float in_p[3], uint8_t out[3])
{
    float ch[3];
    to_channels(in_p, ch);
    out[0] = f32_as_u8(hsv_clamp(ch[0] * 255.0f, 0.0f, 255.0f)); /* :158-162 */
    out[1] = f32_as_u8(hsv_clamp(ch[1] * 255.0f, 0.0f, 255.0f));
    out[2] = f32_as_u8(hsv_clamp(ch[2] * 255.0f, 0.0f, 255.0f));
}

void orc_hsv_to_bgr(const float in_p[3], uint8_t out[3])
{
    float ch[3];
    to_channels(in_p, ch);
    out[0] = f32_as_u8(hsv_clamp(ch[2] * 255.0f, 0.0f, 255.0f)); /* :193-197 */
    out[1] = f32_as_u8(hsv_clamp(ch[1] * 255.0f, 0.0f, 255.0f));
    out[2] = f32_as_u8(hsv_clamp(ch[0] * 255.0f, 0.0f, 255.0f));
}

/* pixel_stride()[0] of the packed formats */
static int bytes_per_pixel(int format)
{
    switch (format) {
    case ORC_FORMAT_RGBX: case ORC_FORMAT_XRGB: case ORC_FORMAT_BGRX: case ORC_FORMAT_XBGR:
    case ORC_FORMAT_RGBA: case ORC_FORMAT_ARGB: case ORC_FORMAT_BGRA: case ORC_FORMAT_ABGR:
        return 4;
    case ORC_FORMAT_RGB: case ORC_FORMAT_BGR:
        return 3;
    default:
        return 0;
    }
}

/* hsvfilter/imp.rs:322-377 picks (offset, rgb|bgr) per format */
static int filter_layout(int format, int *off, int *is_bgr)
{
    switch (format) {
    case ORC_FORMAT_RGBX: case ORC_FORMAT_RGBA: case ORC_FORMAT_RGB: /* :328-338 */
        *off = 0; *is_bgr = 0; return 0;
    case ORC_FORMAT_XRGB: case ORC_FORMAT_ARGB:                      /* :339-349 */
        *off = 1; *is_bgr = 0; return 0;
    case ORC_FORMAT_BGRX: case ORC_FORMAT_BGRA: case ORC_FORMAT_BGR: /* :350-360 */
        *off = 0; *is_bgr = 1; return 0;
    case ORC_FORMAT_XBGR: case ORC_FORMAT_ABGR:                      /* :361-371 */
        *off = 1; *is_bgr = 1; return 0;
    default:
        return -1; /* unreachable!() :372 */
    }
}

/* hsvfilter/imp.rs:76-120 */
int orc_hsvfilter_transform_frame_ip(uint8_t *data, size_t data_len, uint32_t width,
                                     uint32_t stride, int format, const float settings[5])
{
    int off, is_bgr;
    if (filter_layout(format, &off, &is_bgr) != 0)
        return ORC_ERR_FORMAT;
    const size_t nb_channels = (size_t)bytes_per_pixel(format);
    const float hue_shift = settings[0], saturation_mul = settings[1],
                saturation_off = settings[2], value_mul = settings[3], value_off = settings[4];

    if (data_len % nb_channels != 0) /* assert_eq! :92 */
        return ORC_ERR_PANIC;
    if (stride == 0)
        return ORC_ERR_PANIC; /* chunks_exact_mut(0) panics */
    const size_t line_bytes = (size_t)width * nb_channels; /* :94 */
    if (line_bytes > stride)
        return ORC_ERR_PANIC; /* line[..line_bytes] out of range */

    const size_t n_lines = data_len / stride; /* chunks_exact_mut :96 */
    for (size_t y = 0; y < n_lines; y++) {
        uint8_t *line = data + y * (size_t)stride;
        for (size_t xb = 0; xb + nb_channels <= line_bytes; xb += nb_channels) { /* :97 */
            uint8_t *p = line + xb;
            float hsv[3];
            if (is_bgr)
                orc_hsv_from_bgr(p + off, hsv);
            else
                orc_hsv_from_rgb(p + off, hsv);

            hsv[0] = fmodf(hsv[0] + hue_shift, 360.0f); /* :102 */
            if (hsv[0] < 0.0f)                          /* :103-105 */
                hsv[0] += 360.0f;
            hsv[1] = hsv_clamp(saturation_mul * hsv[1] + saturation_off, 0.0f, 1.0f); /* :106-110 */
            hsv[2] = hsv_clamp(value_mul * hsv[2] + value_off, 0.0f, 1.0f);           /* :111-115 */

            uint8_t out[3];
            if (is_bgr)
                orc_hsv_to_bgr(hsv, out);
            else
                orc_hsv_to_rgb(hsv, out);
            p[off + 0] = out[0]; /* :117 apply_filter: 3-byte copy_from_slice */
            p[off + 1] = out[1];
            p[off + 2] = out[2];
        }
    }
    return ORC_OK;
}

/* hsvdetector/imp.rs:422-707: where the true (R,G,B) sit in an input pixel */
static int detect_in_layout(int format, int idx[3], int *is_bgr)
{
    switch (format) {
    case ORC_FORMAT_RGBX: case ORC_FORMAT_RGB: idx[0] = 0; idx[1] = 1; idx[2] = 2; *is_bgr = 0; return 0;
    case ORC_FORMAT_XRGB: idx[0] = 1; idx[1] = 2; idx[2] = 3; *is_bgr = 0; return 0;
    case ORC_FORMAT_BGRX: case ORC_FORMAT_BGR: idx[0] = 2; idx[1] = 1; idx[2] = 0; *is_bgr = 1; return 0;
    case ORC_FORMAT_XBGR: idx[0] = 3; idx[1] = 2; idx[2] = 1; *is_bgr = 1; return 0;
    default: return -1;
    }
}

/* where (R,G,B,alpha) go in an output pixel */
static int detect_out_layout(int format, int idx[4])
{
    switch (format) {
    case ORC_FORMAT_RGBA: idx[0] = 0; idx[1] = 1; idx[2] = 2; idx[3] = 3; return 0;
    case ORC_FORMAT_ARGB: idx[0] = 1; idx[1] = 2; idx[2] = 3; idx[3] = 0; return 0;
    case ORC_FORMAT_BGRA: idx[0] = 2; idx[1] = 1; idx[2] = 0; idx[3] = 3; return 0;
    case ORC_FORMAT_ABGR: idx[0] = 3; idx[1] = 2; idx[2] = 1; idx[3] = 0; return 0;
    default: return -1;
    }
}

/* hsvdetector/imp.rs:100-160 */
int orc_hsvdetector_transform_frame(const uint8_t *in_data, size_t in_len, uint32_t in_stride,
                                    int in_format, uint8_t *out_data, size_t out_len,
                                    uint32_t out_stride, int out_format, uint32_t width,
                                    const float settings[6])
{
    int in_idx[3], out_idx[4], is_bgr;
    if (detect_in_layout(in_format, in_idx, &is_bgr) != 0 ||
        detect_out_layout(out_format, out_idx) != 0)
        return ORC_ERR_FORMAT;
    const size_t nb_in = (size_t)bytes_per_pixel(in_format);
    const float hue_ref = settings[0], hue_var = settings[1], saturation_ref = settings[2],
                saturation_var = settings[3], value_ref = settings[4], value_var = settings[5];

    if (in_stride == 0 || out_stride == 0)
        return ORC_ERR_PANIC;
    if (out_len / out_stride != in_len / in_stride) /* assert_eq! :121 */
        return ORC_ERR_PANIC;
    if (in_len % nb_in != 0) /* assert_eq! :122 */
        return ORC_ERR_PANIC;
    const size_t in_line_bytes = (size_t)width * nb_in; /* :124-125 */
    const size_t out_line_bytes = (size_t)width * 4;
    if (in_line_bytes > in_stride || out_line_bytes > out_stride) /* :127-128 */
        return ORC_ERR_PANIC;

    const size_t n_lines = in_len / in_stride;
    for (size_t y = 0; y < n_lines; y++) { /* :130-133 */
        const uint8_t *in_line = in_data + y * (size_t)in_stride;
        uint8_t *out_line = out_data + y * (size_t)out_stride;
        for (size_t x = 0; x < width; x++) { /* :134-137 */
            const uint8_t *in_p = in_line + x * nb_in;
            uint8_t *out_p = out_line + x * 4;
            /* from_rgb on the RGB-ordered bytes / from_bgr on the BGR-ordered bytes: both
             * see the true (R,G,B); from_bgr reads p[0]=b,p[2]=r (hsvutils.rs:89-91) */
            uint8_t first3[3];
            float hsv[3];
            if (is_bgr) {
                first3[0] = in_p[in_idx[2]]; first3[1] = in_p[in_idx[1]]; first3[2] = in_p[in_idx[0]];
                orc_hsv_from_bgr(first3, hsv);
            } else {
                first3[0] = in_p[in_idx[0]]; first3[1] = in_p[in_idx[1]]; first3[2] = in_p[in_idx[2]];
                orc_hsv_from_rgb(first3, hsv);
            }

            float ref_hue_offset = 180.0f - hue_ref;    /* :141 */
            float shifted_hue = hsv[0] + ref_hue_offset; /* :142 */
            if (shifted_hue < 0.0f)                      /* :144-146 */
                shifted_hue += 360.0f;
            shifted_hue = fmodf(shifted_hue, 360.0f);    /* :148 */

            uint8_t alpha;
            if (fabsf(shifted_hue - 180.0f) <= hue_var &&      /* :150-152 */
                fabsf(hsv[1] - saturation_ref) <= saturation_var &&
                fabsf(hsv[2] - value_ref) <= value_var)
                alpha = 255;
            else
                alpha = 0;

            out_p[out_idx[0]] = in_p[in_idx[0]]; /* apply_alpha closures :428-704 */
            out_p[out_idx[1]] = in_p[in_idx[1]];
            out_p[out_idx[2]] = in_p[in_idx[2]];
            out_p[out_idx[3]] = alpha;
        }
    }
    return ORC_OK;
}

void orc_hsv_from_rgb_frame(const uint8_t *rgbx, size_t n_pixels, float *hsv_out)
{
    for (size_t i = 0; i < n_pixels; i++)
        orc_hsv_from_rgb(rgbx + 4 * i, hsv_out + 3 * i);
}
