/*
 * oracle/ssim_oracle.c -- CPU restatement (f64) of the SSIM-family distance behind
 * videocompare's `hash-algo=dssim` (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED.  The reference delegates to the crate dssim-core 3.4.0 (Cargo.lock:3625-3634,
 * call sites video/videofx/src/videocompare/hashed_image.rs:49-59,72-75) behind the NON-DEFAULT
 * cargo feature `dssim` (video/videofx/Cargo.toml:39); its source is not under /root/reference.
 * What is restated here is the published structure of that algorithm as recorded in
 * SURVEY.md Appendix A.3 (sRGB -> linear -> Lab-like planes, 5-level 2x box pyramid, binomial blur,
 * per-scale SSIM map, mean adjusted by mean absolute deviation, fixed scale weights, 1/ssim - 1).
 * Constants that SURVEY.md does not fix are chosen here and documented inline; the numeric value
 * is therefore NOT expected to equal dssim-core's.  What the reference's own test pins
 * (tests/videocompare.rs:141-182: identical frames => distance <= 0) holds exactly: identical
 * inputs give 0.0.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SSIM_SCALES 5
static const double SCALE_WEIGHTS[SSIM_SCALES] = {0.028, 0.197, 0.322, 0.298, 0.155}; /* SURVEY A.3 */
static const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;

static double srgb_to_linear(int b)
{
    const double x = b / 255.0;
    return x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
}

static double lab_f(double t)
{
    const double eps = 216.0 / 24389.0, kappa = 24389.0 / 27.0;
    return t > eps ? cbrt(t) : (kappa * t + 16.0) / 116.0;
}

/* linear RGB -> the three planes the SSIM runs on, each roughly in [0,1] */
static void to_lab(double r, double g, double b, double out[3])
{
    const double X = (0.4124 * r + 0.3576 * g + 0.1805 * b) / 0.9505;
    const double Y = 0.2126 * r + 0.7152 * g + 0.0722 * b;
    const double Z = (0.0193 * r + 0.1192 * g + 0.9505 * b) / 1.089;
    const double fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    out[0] = (116.0 * fy - 16.0) / 100.0;
    out[1] = (86.2 + 500.0 * (fx - fy)) / 220.0;
    out[2] = (107.9 + 200.0 * (fy - fz)) / 220.0;
}

typedef struct { int w, h; double *p[3]; } planes_t;

static planes_t planes_new(int w, int h)
{
    planes_t q;
    q.w = w; q.h = h;
    for (int c = 0; c < 3; c++) q.p[c] = malloc(sizeof(double) * (size_t)w * h);
    return q;
}
static void planes_free(planes_t *q) { for (int c = 0; c < 3; c++) free(q->p[c]); }

/* frame -> linear RGB planes, alpha premultiplied (RGBA) */
static planes_t linearize(const uint8_t *data, uint32_t w, uint32_t h, uint32_t stride, int bpp)
{
    double lut[256];
    for (int i = 0; i < 256; i++) lut[i] = srgb_to_linear(i);
    planes_t q = planes_new((int)w, (int)h);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *p = data + (size_t)y * stride + (size_t)x * bpp;
            const double a = bpp == 4 ? p[3] / 255.0 : 1.0;
            for (int c = 0; c < 3; c++) q.p[c][(size_t)y * w + x] = lut[p[c]] * a;
        }
    return q;
}

static planes_t downsample(const planes_t *s)
{
    planes_t q = planes_new(s->w / 2, s->h / 2);
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < q.h; y++)
            for (int x = 0; x < q.w; x++) {
                const double *r0 = s->p[c] + (size_t)(2 * y) * s->w + 2 * x, *r1 = r0 + s->w;
                q.p[c][(size_t)y * q.w + x] = (r0[0] + r0[1] + r1[0] + r1[1]) * 0.25;
            }
    return q;
}

static planes_t lab_of(const planes_t *lin)
{
    planes_t q = planes_new(lin->w, lin->h);
    const size_t n = (size_t)lin->w * lin->h;
    for (size_t i = 0; i < n; i++) {
        double o[3];
        to_lab(lin->p[0][i], lin->p[1][i], lin->p[2][i], o);
        q.p[0][i] = o[0]; q.p[1][i] = o[1]; q.p[2][i] = o[2];
    }
    return q;
}

static const double BINOM[5] = {1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* SSIM map of one scale (mean of the three channels) */
static void ssim_map(const planes_t *a, const planes_t *b, double *map)
{
    const int w = a->w, h = a->h;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double acc = 0.0;
            for (int c = 0; c < 3; c++) {
                double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
                for (int dy = -2; dy <= 2; dy++) {
                    const int yy = clampi(y + dy, 0, h - 1);
                    for (int dx = -2; dx <= 2; dx++) {
                        const int xx = clampi(x + dx, 0, w - 1);
                        const double wgt = BINOM[dy + 2] * BINOM[dx + 2];
                        const double v1 = a->p[c][(size_t)yy * w + xx], v2 = b->p[c][(size_t)yy * w + xx];
                        m1 += wgt * v1; m2 += wgt * v2;
                        s11 += wgt * v1 * v1; s22 += wgt * v2 * v2; s12 += wgt * v1 * v2;
                    }
                }
                s11 -= m1 * m1; s22 -= m2 * m2; s12 -= m1 * m2;
                acc += ((2.0 * m1 * m2 + C1) * (2.0 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s11 + s22 + C2));
            }
            map[(size_t)y * w + x] = acc / 3.0;
        }
}

/* Returns the distance (0 = identical, grows with dissimilarity); per_scale (optional, 5 doubles)
 * receives mean - MAD of each scale (NaN for scales that were skipped because < 8 px). */
int orc_ssim_distance(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t stride_a,
                      uint32_t stride_b, int format, double *distance, double *per_scale)
{
    int bpp;
    if (format == ORC_FORMAT_RGB) bpp = 3;
    else if (format == ORC_FORMAT_RGBA) bpp = 4;
    else return ORC_ERR_FORMAT;
    if (width < 8 || height < 8) return ORC_ERR_PANIC;
    planes_t la = linearize(a, width, height, stride_a, bpp), lb = linearize(b, width, height, stride_b, bpp);
    double num = 0.0, den = 0.0;
    for (int s = 0; s < SSIM_SCALES; s++) {
        if (per_scale) per_scale[s] = NAN;
        if (s > 0) {
            if (la.w / 2 < 8 || la.h / 2 < 8) break;
            planes_t na = downsample(&la), nb = downsample(&lb);
            planes_free(&la); planes_free(&lb);
            la = na; lb = nb;
        }
        planes_t A = lab_of(&la), B = lab_of(&lb);
        const size_t n = (size_t)la.w * la.h;
        double *map = malloc(sizeof(double) * n);
        ssim_map(&A, &B, map);
        double sum = 0.0;
        for (size_t i = 0; i < n; i++) sum += map[i];
        const double avg = sum / (double)n;
        double dev = 0.0;
        for (size_t i = 0; i < n; i++) dev += fabs(map[i] - avg);
        const double score = avg - dev / (double)n;
        if (per_scale) per_scale[s] = score;
        num += SCALE_WEIGHTS[s] * score;
        den += SCALE_WEIGHTS[s];
        free(map);
        planes_free(&A); planes_free(&B);
    }
    planes_free(&la); planes_free(&lb);
    const double ssim = num / den;
    *distance = 1.0 / (ssim > 1e-12 ? ssim : 1e-12) - 1.0;
    return ORC_OK;
}

/* Row-band partial of the same computation, for the world_size-2 tests of the sharded path:
 * mean == NULL -> sums[s] = sum of the scale-s map over the band's rows, counts[s] = pixels;
 * mean != NULL -> sums[s] = sum of |map - mean[s]| over the same rows.
 * Band rows at scale s are [row_begin >> s, row_end >> s) (row_end == height -> to the bottom). */
int orc_ssim_band(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t stride_a,
                  uint32_t stride_b, int format, uint32_t row_begin, uint32_t row_end, const double *mean,
                  double *sums, double *counts, int *n_scales)
{
    int bpp;
    if (format == ORC_FORMAT_RGB) bpp = 3;
    else if (format == ORC_FORMAT_RGBA) bpp = 4;
    else return ORC_ERR_FORMAT;
    if (width < 8 || height < 8) return ORC_ERR_PANIC;
    planes_t la = linearize(a, width, height, stride_a, bpp), lb = linearize(b, width, height, stride_b, bpp);
    *n_scales = 0;
    for (int s = 0; s < SSIM_SCALES; s++) {
        sums[s] = 0.0; counts[s] = 0.0;
        if (s > 0 && (la.w / 2 < 8 || la.h / 2 < 8)) { for (int t = s; t < SSIM_SCALES; t++) { sums[t] = 0; counts[t] = 0; } break; }
        if (s > 0) {
            planes_t na = downsample(&la), nb = downsample(&lb);
            planes_free(&la); planes_free(&lb);
            la = na; lb = nb;
        }
        planes_t A = lab_of(&la), B = lab_of(&lb);
        const size_t n = (size_t)la.w * la.h;
        double *map = malloc(sizeof(double) * n);
        ssim_map(&A, &B, map);
        const int y0 = (int)(row_begin >> s);
        int y1 = row_end >= height ? la.h : (int)(row_end >> s);
        if (y1 > la.h) y1 = la.h;
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < la.w; x++) {
                const double v = map[(size_t)y * la.w + x];
                sums[s] += mean ? fabs(v - mean[s]) : v;
            }
        counts[s] = y1 > y0 ? (double)(y1 - y0) * la.w : 0.0;
        *n_scales = s + 1;
        free(map);
        planes_free(&A); planes_free(&B);
    }
    planes_free(&la); planes_free(&lb);
    return ORC_OK;
}

double orc_ssim_combine(const double *mean, const double *mad, int n_scales)
{
    double num = 0.0, den = 0.0;
    for (int s = 0; s < n_scales && s < SSIM_SCALES; s++) {
        num += SCALE_WEIGHTS[s] * (mean[s] - mad[s]);
        den += SCALE_WEIGHTS[s];
    }
    const double ssim = den > 0 ? num / den : 1.0;
    return 1.0 / (ssim > 1e-12 ? ssim : 1e-12) - 1.0;
}
