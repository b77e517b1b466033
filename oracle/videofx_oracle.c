/*
 * oracle/videofx_oracle.c -- CPU restatement of the third-party algorithms behind
 * video/videofx colordetect and videocompare (TEST INFRASTRUCTURE ONLY).
 *
 * The arithmetic is NOT under /root/reference: it lives in crates pinned by Cargo.lock
 *   color-thief 0.2.2 (Cargo.lock:2051-2056)  call site colordetect/imp.rs:68-74
 *   color-name  1.2.0 (Cargo.lock:2045-2048)  call site colordetect/imp.rs:77-79
 *   image_hasher 3.1.1 (Cargo.lock:7459-7468) call sites videocompare/hashed_image.rs:37-45,70
 * This file restates their published algorithms (MMCQ modified median cut from Leptonica ->
 * quantize.js -> color-thief; the "blockhash" perceptual hash, integer fast path and f32 slow path).
 * PARITY UNPINNED beyond what the reference's own pipeline tests pin:
 *   video/videofx/tests/colordetect.rs:21-68   solid red  => dominant-color "red"
 *   video/videofx/tests/videocompare.rs:57-139 red vs red => distance 0; snow vs red => > 0
 * Outputs generated from this file are labelled "self-golden" in tests/golden.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ colordetect */

#define SIGNAL_BITS 5
#define RIGHT_SHIFT (8 - SIGNAL_BITS)
#define MULTIPLIER (1 << RIGHT_SHIFT)
#define HISTOGRAM_SIZE (1 << (3 * SIGNAL_BITS))
#define VBOX_LENGTH (1 << SIGNAL_BITS)
#define FRACTION_BY_POPULATION 0.75
#define MAX_ITERATIONS 1000

static inline int color_index(int r, int g, int b) { return (r << (2 * SIGNAL_BITS)) + (g << SIGNAL_BITS) + b; }

/* colordetect/imp.rs:260-283 maps the negotiated format to color_thief::ColorFormat */
static int cd_layout(int format, int *bpp, int idx[4])
{
    switch (format) {
    case ORC_FORMAT_RGB:  *bpp = 3; idx[0] = 0; idx[1] = 1; idx[2] = 2; idx[3] = -1; return 0;
    case ORC_FORMAT_RGBA: *bpp = 4; idx[0] = 0; idx[1] = 1; idx[2] = 2; idx[3] = 3; return 0;
    case ORC_FORMAT_ARGB: *bpp = 4; idx[0] = 1; idx[1] = 2; idx[2] = 3; idx[3] = 0; return 0;
    case ORC_FORMAT_BGR:  *bpp = 3; idx[0] = 2; idx[1] = 1; idx[2] = 0; idx[3] = -1; return 0;
    case ORC_FORMAT_BGRA: *bpp = 4; idx[0] = 2; idx[1] = 1; idx[2] = 0; idx[3] = 3; return 0;
    default: return -1;
    }
}

/* color-thief make_histogram_and_vbox: every quality-th pixel of the FLAT byte slice (the
 * element passes the whole plane incl. row padding, colordetect/imp.rs:69) */
int orc_colordetect_histogram(const uint8_t *pixels, size_t len, int format, uint32_t quality,
                              int32_t *hist, uint32_t minmax[6], uint64_t *n_counted)
{
    int bpp, idx[4];
    if (cd_layout(format, &bpp, idx) != 0)
        return ORC_ERR_FORMAT;
    if (quality < 1 || quality > 10)
        return ORC_ERR_PANIC; /* assert!(quality > 0 && quality <= 10) */
    memset(hist, 0, HISTOGRAM_SIZE * sizeof(int32_t));
    uint32_t rmin = 255, rmax = 0, gmin = 255, gmax = 0, bmin = 255, bmax = 0;
    uint64_t counted = 0;
    size_t pixel_count = len / (size_t)bpp;
    for (size_t i = 0; i < pixel_count; i += quality) {
        const uint8_t *p = pixels + i * (size_t)bpp;
        uint32_t r = p[idx[0]], g = p[idx[1]], b = p[idx[2]];
        uint32_t a = idx[3] >= 0 ? p[idx[3]] : 255;
        if (a < 125 || (r > 250 && g > 250 && b > 250))
            continue;
        r >>= RIGHT_SHIFT; g >>= RIGHT_SHIFT; b >>= RIGHT_SHIFT;
        if (r < rmin) rmin = r;
        if (r > rmax) rmax = r;
        if (g < gmin) gmin = g;
        if (g > gmax) gmax = g;
        if (b < bmin) bmin = b;
        if (b > bmax) bmax = b;
        hist[color_index((int)r, (int)g, (int)b)] += 1;
        counted++;
    }
    minmax[0] = rmin; minmax[1] = rmax; minmax[2] = gmin;
    minmax[3] = gmax; minmax[4] = bmin; minmax[5] = bmax;
    if (n_counted) *n_counted = counted;
    return ORC_OK;
}

typedef struct {
    int r_min, r_max, g_min, g_max, b_min, b_max;
    int avg[3];
    int volume;
    int count;
} vbox_t;

static void vbox_recalc(vbox_t *v, const int32_t *hist)
{
    /* i32 accumulators as in the crate; a release build of the reference wraps on overflow (first possible at
     * ~8.5 M samples in one box, i.e. 8K frames at quality <= 3), so the sums are carried as u32 and read back as i32 */
    uint32_t ntot_u = 0, r_u = 0, g_u = 0, b_u = 0, count_u = 0;
    for (int i = v->r_min; i <= v->r_max; i++)
        for (int j = v->g_min; j <= v->g_max; j++)
            for (int k = v->b_min; k <= v->b_max; k++) {
                double hval = (double)hist[color_index(i, j, k)];
                ntot_u += (uint32_t)(int32_t)hval;
                r_u += (uint32_t)(int32_t)(hval * ((double)i + 0.5) * (double)MULTIPLIER);
                g_u += (uint32_t)(int32_t)(hval * ((double)j + 0.5) * (double)MULTIPLIER);
                b_u += (uint32_t)(int32_t)(hval * ((double)k + 0.5) * (double)MULTIPLIER);
                count_u += (uint32_t)hist[color_index(i, j, k)];
            }
    const int ntot = (int)(int32_t)ntot_u, r_sum = (int)(int32_t)r_u, g_sum = (int)(int32_t)g_u, b_sum = (int)(int32_t)b_u,
              count = (int)(int32_t)count_u;
    if (ntot > 0) {
        v->avg[0] = (r_sum / ntot) & 0xff; /* `as u8` on i32 wraps */
        v->avg[1] = (g_sum / ntot) & 0xff;
        v->avg[2] = (b_sum / ntot) & 0xff;
    } else {
        int r = MULTIPLIER * (v->r_min + v->r_max + 1) / 2;
        int g = MULTIPLIER * (v->g_min + v->g_max + 1) / 2;
        int b = MULTIPLIER * (v->b_min + v->b_max + 1) / 2;
        v->avg[0] = r < 255 ? r : 255;
        v->avg[1] = g < 255 ? g : 255;
        v->avg[2] = b < 255 ? b : 255;
    }
    v->count = count;
    v->volume = (v->r_max - v->r_min + 1) * (v->g_max - v->g_min + 1) * (v->b_max - v->b_min + 1);
}

static int cmp_count(const vbox_t *a, const vbox_t *b) { return (a->count > b->count) - (a->count < b->count); }

static int cmp_product(const vbox_t *a, const vbox_t *b)
{
    if (a->count == b->count)
        return (a->volume > b->volume) - (a->volume < b->volume);
    int64_t pa = (int64_t)a->count * a->volume, pb = (int64_t)b->count * b->volume;
    return (pa > pb) - (pa < pb);
}

/* Vec::sort_by is a stable sort: insertion sort keeps equal elements in order */
static void stable_sort(vbox_t *q, int n, int (*cmp)(const vbox_t *, const vbox_t *))
{
    for (int i = 1; i < n; i++) {
        vbox_t key = q[i];
        int j = i - 1;
        while (j >= 0 && cmp(&q[j], &key) > 0) { q[j + 1] = q[j]; j--; }
        q[j + 1] = key;
    }
}

/* apply_median_cut + cut; returns number of boxes produced (1 or 2) or <0 */
static int median_cut(const int32_t *hist, const vbox_t *vbox, vbox_t out[2])
{
    if (vbox->count == 0)
        return -1;
    if (vbox->count == 1) { out[0] = *vbox; return 1; }

    int rw = vbox->r_max - vbox->r_min, gw = vbox->g_max - vbox->g_min, bw = vbox->b_max - vbox->b_min;
    int mx = rw > gw ? rw : gw; if (bw > mx) mx = bw;
    int axis = (mx == rw) ? 0 : (mx == gw) ? 1 : 2;

    int partial[VBOX_LENGTH], look_ahead[VBOX_LENGTH];
    for (int i = 0; i < VBOX_LENGTH; i++) { partial[i] = -1; look_ahead[i] = -1; }
    int total = 0;
    int lo[3] = {vbox->r_min, vbox->g_min, vbox->b_min}, hi[3] = {vbox->r_max, vbox->g_max, vbox->b_max};
    for (int i = lo[axis]; i <= hi[axis]; i++) {
        int sum = 0;
        int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
        for (int j = lo[a1]; j <= hi[a1]; j++)
            for (int k = lo[a2]; k <= hi[a2]; k++) {
                int c[3];
                c[axis] = i; c[a1] = j; c[a2] = k;
                sum += hist[color_index(c[0], c[1], c[2])];
            }
        total += sum;
        partial[i] = total;
    }
    for (int i = 0; i < VBOX_LENGTH; i++)
        if (partial[i] != -1)
            look_ahead[i] = total - partial[i];

    int vmin = lo[axis], vmax = hi[axis];
    for (int i = vmin; i <= vmax; i++) {
        if (partial[i] > total / 2) {
            vbox_t v1 = *vbox, v2 = *vbox;
            int left = i - vmin, right = vmax - i;
            int d2;
            if (left <= right) {
                int t = i + right / 2;
                d2 = (vmax - 1 < t) ? vmax - 1 : t;
            } else {
                int t = (int)((double)(i - 1) - (double)left / 2.0);
                d2 = (vmin > t) ? vmin : t;
            }
            while (d2 < 0 || partial[d2] <= 0)
                d2++;
            int count2 = look_ahead[d2];
            while (count2 == 0 && d2 > 0 && partial[d2 - 1] > 0) {
                d2--;
                count2 = look_ahead[d2];
            }
            if (axis == 0) { v1.r_max = d2; v2.r_min = d2 + 1; }
            else if (axis == 1) { v1.g_max = d2; v2.g_min = d2 + 1; }
            else { v1.b_max = d2; v2.b_min = d2 + 1; }
            vbox_recalc(&v1, hist);
            vbox_recalc(&v2, hist);
            out[0] = v1; out[1] = v2;
            return 2;
        }
    }
    return -2;
}

static int mmcq_iterate(vbox_t *q, int *n, int (*cmp)(const vbox_t *, const vbox_t *), int target,
                        const int32_t *hist)
{
    int color = 1;
    for (int it = 0; it < MAX_ITERATIONS; it++) {
        if (*n == 0)
            break;
        vbox_t vbox = q[*n - 1];
        if (vbox.count == 0) {
            stable_sort(q, *n, cmp);
            continue;
        }
        (*n)--;
        vbox_t res[2];
        int k = median_cut(hist, &vbox, res);
        if (k < 0)
            return k;
        q[(*n)++] = res[0];
        if (k == 2) {
            q[(*n)++] = res[1];
            color++;
        }
        stable_sort(q, *n, cmp);
        if (color >= target)
            break;
    }
    return 0;
}

int orc_mmcq_from_histogram(const int32_t *hist, const uint32_t minmax[6], uint32_t max_colors,
                            uint32_t *palette_out)
{
    if (max_colors < 2 || max_colors > 255)
        return ORC_ERR_PANIC; /* assert!(max_colors > 1); u8 argument */
    vbox_t q[600];
    int n = 0;
    vbox_t v0;
    v0.r_min = (int)minmax[0]; v0.r_max = (int)minmax[1];
    v0.g_min = (int)minmax[2]; v0.g_max = (int)minmax[3];
    v0.b_min = (int)minmax[4]; v0.b_max = (int)minmax[5];
    vbox_recalc(&v0, hist);
    q[n++] = v0;

    int target = (int)ceil(FRACTION_BY_POPULATION * (double)max_colors);
    int rc = mmcq_iterate(q, &n, cmp_count, target, hist);
    if (rc < 0) return rc;
    stable_sort(q, n, cmp_product);
    rc = mmcq_iterate(q, &n, cmp_product, (int)max_colors - n, hist);
    if (rc < 0) return rc;

    int outn = 0;
    for (int i = n - 1; i >= 0 && outn < (int)max_colors; i--) { /* reverse + truncate */
        palette_out[outn++] = ((uint32_t)q[i].avg[0] << 16) | ((uint32_t)q[i].avg[1] << 8) | (uint32_t)q[i].avg[2];
    }
    return outn;
}

int orc_colordetect_palette(const uint8_t *pixels, size_t len, int format, uint32_t quality,
                            uint32_t max_colors, uint32_t *palette_out)
{
    int32_t *hist = malloc(HISTOGRAM_SIZE * sizeof(int32_t));
    uint32_t minmax[6];
    int rc = orc_colordetect_histogram(pixels, len, format, quality, hist, minmax, NULL);
    if (rc == ORC_OK)
        rc = orc_mmcq_from_histogram(hist, minmax, max_colors, palette_out);
    free(hist);
    return rc;
}

/* CSS Color Module Level 4 named colours, alphabetical (color-name's css table). */
static const struct { const char *name; uint8_t r, g, b; } CSS_COLORS[] = {
    {"aliceblue", 240, 248, 255}, {"antiquewhite", 250, 235, 215}, {"aqua", 0, 255, 255},
    {"aquamarine", 127, 255, 212}, {"azure", 240, 255, 255}, {"beige", 245, 245, 220},
    {"bisque", 255, 228, 196}, {"black", 0, 0, 0}, {"blanchedalmond", 255, 235, 205},
    {"blue", 0, 0, 255}, {"blueviolet", 138, 43, 226}, {"brown", 165, 42, 42},
    {"burlywood", 222, 184, 135}, {"cadetblue", 95, 158, 160}, {"chartreuse", 127, 255, 0},
    {"chocolate", 210, 105, 30}, {"coral", 255, 127, 80}, {"cornflowerblue", 100, 149, 237},
    {"cornsilk", 255, 248, 220}, {"crimson", 220, 20, 60}, {"cyan", 0, 255, 255},
    {"darkblue", 0, 0, 139}, {"darkcyan", 0, 139, 139}, {"darkgoldenrod", 184, 134, 11},
    {"darkgray", 169, 169, 169}, {"darkgreen", 0, 100, 0}, {"darkgrey", 169, 169, 169},
    {"darkkhaki", 189, 183, 107}, {"darkmagenta", 139, 0, 139}, {"darkolivegreen", 85, 107, 47},
    {"darkorange", 255, 140, 0}, {"darkorchid", 153, 50, 204}, {"darkred", 139, 0, 0},
    {"darksalmon", 233, 150, 122}, {"darkseagreen", 143, 188, 143}, {"darkslateblue", 72, 61, 139},
    {"darkslategray", 47, 79, 79}, {"darkslategrey", 47, 79, 79}, {"darkturquoise", 0, 206, 209},
    {"darkviolet", 148, 0, 211}, {"deeppink", 255, 20, 147}, {"deepskyblue", 0, 191, 255},
    {"dimgray", 105, 105, 105}, {"dimgrey", 105, 105, 105}, {"dodgerblue", 30, 144, 255},
    {"firebrick", 178, 34, 34}, {"floralwhite", 255, 250, 240}, {"forestgreen", 34, 139, 34},
    {"fuchsia", 255, 0, 255}, {"gainsboro", 220, 220, 220}, {"ghostwhite", 248, 248, 255},
    {"gold", 255, 215, 0}, {"goldenrod", 218, 165, 32}, {"gray", 128, 128, 128},
    {"green", 0, 128, 0}, {"greenyellow", 173, 255, 47}, {"grey", 128, 128, 128},
    {"honeydew", 240, 255, 240}, {"hotpink", 255, 105, 180}, {"indianred", 205, 92, 92},
    {"indigo", 75, 0, 130}, {"ivory", 255, 255, 240}, {"khaki", 240, 230, 140},
    {"lavender", 230, 230, 250}, {"lavenderblush", 255, 240, 245}, {"lawngreen", 124, 252, 0},
    {"lemonchiffon", 255, 250, 205}, {"lightblue", 173, 216, 230}, {"lightcoral", 240, 128, 128},
    {"lightcyan", 224, 255, 255}, {"lightgoldenrodyellow", 250, 250, 210}, {"lightgray", 211, 211, 211},
    {"lightgreen", 144, 238, 144}, {"lightgrey", 211, 211, 211}, {"lightpink", 255, 182, 193},
    {"lightsalmon", 255, 160, 122}, {"lightseagreen", 32, 178, 170}, {"lightskyblue", 135, 206, 250},
    {"lightslategray", 119, 136, 153}, {"lightslategrey", 119, 136, 153}, {"lightsteelblue", 176, 196, 222},
    {"lightyellow", 255, 255, 224}, {"lime", 0, 255, 0}, {"limegreen", 50, 205, 50},
    {"linen", 250, 240, 230}, {"magenta", 255, 0, 255}, {"maroon", 128, 0, 0},
    {"mediumaquamarine", 102, 205, 170}, {"mediumblue", 0, 0, 205}, {"mediumorchid", 186, 85, 211},
    {"mediumpurple", 147, 112, 219}, {"mediumseagreen", 60, 179, 113}, {"mediumslateblue", 123, 104, 238},
    {"mediumspringgreen", 0, 250, 154}, {"mediumturquoise", 72, 209, 204}, {"mediumvioletred", 199, 21, 133},
    {"midnightblue", 25, 25, 112}, {"mintcream", 245, 255, 250}, {"mistyrose", 255, 228, 225},
    {"moccasin", 255, 228, 181}, {"navajowhite", 255, 222, 173}, {"navy", 0, 0, 128},
    {"oldlace", 253, 245, 230}, {"olive", 128, 128, 0}, {"olivedrab", 107, 142, 35},
    {"orange", 255, 165, 0}, {"orangered", 255, 69, 0}, {"orchid", 218, 112, 214},
    {"palegoldenrod", 238, 232, 170}, {"palegreen", 152, 251, 152}, {"paleturquoise", 175, 238, 238},
    {"palevioletred", 219, 112, 147}, {"papayawhip", 255, 239, 213}, {"peachpuff", 255, 218, 185},
    {"peru", 205, 133, 63}, {"pink", 255, 192, 203}, {"plum", 221, 160, 221},
    {"powderblue", 176, 224, 230}, {"purple", 128, 0, 128}, {"rebeccapurple", 102, 51, 153},
    {"red", 255, 0, 0}, {"rosybrown", 188, 143, 143}, {"royalblue", 65, 105, 225},
    {"saddlebrown", 139, 69, 19}, {"salmon", 250, 128, 114}, {"sandybrown", 244, 164, 96},
    {"seagreen", 46, 139, 87}, {"seashell", 255, 245, 238}, {"sienna", 160, 82, 45},
    {"silver", 192, 192, 192}, {"skyblue", 135, 206, 235}, {"slateblue", 106, 90, 205},
    {"slategray", 112, 128, 144}, {"slategrey", 112, 128, 144}, {"snow", 255, 250, 250},
    {"springgreen", 0, 255, 127}, {"steelblue", 70, 130, 180}, {"tan", 210, 180, 140},
    {"teal", 0, 128, 128}, {"thistle", 216, 191, 216}, {"tomato", 255, 99, 71},
    {"turquoise", 64, 224, 208}, {"violet", 238, 130, 238}, {"wheat", 245, 222, 179},
    {"white", 255, 255, 255}, {"whitesmoke", 245, 245, 245}, {"yellow", 255, 255, 0},
    {"yellowgreen", 154, 205, 50},
};

/* nearest CSS colour by squared Euclidean RGB distance; first minimum wins (tie-break
 * of the crate is unknown => "parity unpinned") */
const char *orc_css_color_similar(uint8_t r, uint8_t g, uint8_t b)
{
    size_t best = 0;
    long best_d = -1;
    for (size_t i = 0; i < sizeof(CSS_COLORS) / sizeof(CSS_COLORS[0]); i++) {
        long dr = (long)r - CSS_COLORS[i].r, dg = (long)g - CSS_COLORS[i].g, db = (long)b - CSS_COLORS[i].b;
        long d = dr * dr + dg * dg + db * db;
        if (best_d < 0 || d < best_d) { best_d = d; best = i; }
    }
    return CSS_COLORS[best].name;
}

/* ------------------------------------------------------------------ videocompare */

/* image_hasher 3.1.1 `blockhash(img, 8, 8)` (src/alg/blockhash.rs; crate not under /root/reference: PARITY UNPINNED,
 * restated from the crate's published source).  Pixel value r+g+b, RGBA: 765 if a==0 (`sum_px`).  The frame is first
 * tightly packed (hashed_image.rs:110-130): row padding never counts.
 *
 *   W%8==0 && H%8==0  -> `blockhash_fast`: u32 block sums over W/8 x H/8 blocks.
 *   otherwise         -> `blockhash_slow`: f32 block sums, every pixel visited in raster order:
 *        block_width = W as f32 / 8.0; block_x = x / block_width; x_mod = x + 1. % block_width   (sic: `%` binds tighter
 *        than `+`, so x_mod = x + fmod(1, block_width), which is x + 1 whenever W > 8);
 *        weight_left = fract(x_mod), weight_right = 1 - weight_left; block_left = floor(block_x);
 *        block_right = trunc(x_mod) == 0 ? ceil(block_x) : block_left   (same in y with top/bottom);
 *        blocks[top][left] += p*wl*wt; blocks[bottom][left] += p*wl*wb; blocks[top][right] += p*wr*wt;
 *        blocks[bottom][right] += p*wr*wb       -- four f32 `+=` per pixel, in this order.
 *      (trunc(x_mod) == 0 only for x == 0, where ceil(block_x) == 0 == block_left, so right == left and bottom == top
 *      always; for W > 8 and H > 8 the weights are exactly 0/1 and a pixel adds `p` once and +0.0 three times.)
 * The 64 sums are returned as 32-bit words: u32 values on the fast path, f32 BIT PATTERNS on the slow path. */
static int blockhash_is_fast(uint32_t width, uint32_t height) { return width % 8 == 0 && height % 8 == 0; }

int orc_blockhash_sums(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride,
                       int format, uint32_t sums[64])
{
    int bpp;
    if (format == ORC_FORMAT_RGB) bpp = 3;
    else if (format == ORC_FORMAT_RGBA) bpp = 4;
    else return ORC_ERR_FORMAT; /* videocompare caps: RGB, RGBA only (imp.rs:160-162) */
    if (width == 0 || height == 0)
        return ORC_ERR_PANIC;
    memset(sums, 0, 64 * sizeof(uint32_t));
    if (blockhash_is_fast(width, height)) { /* blockhash_fast */
        uint32_t bw = width / 8, bh = height / 8;
        for (uint32_t y = 0; y < height; y++) {
            const uint8_t *row = data + (size_t)y * stride;
            for (uint32_t x = 0; x < width; x++) {
                const uint8_t *p = row + (size_t)x * (size_t)bpp;
                uint32_t v = (uint32_t)p[0] + p[1] + p[2];
                if (bpp == 4 && p[3] == 0)
                    v = 765;
                sums[(y / bh) * 8 + x / bw] += v;
            }
        }
        return ORC_OK;
    }
    /* blockhash_slow */
    float blocks[64];
    for (int i = 0; i < 64; i++) blocks[i] = 0.0f;
    const float block_width = (float)width / 8.0f, block_height = (float)height / 8.0f;
    const float mx = fmodf(1.0f, block_width), my = fmodf(1.0f, block_height);
    for (uint32_t yi = 0; yi < height; yi++) {
        const uint8_t *row = data + (size_t)yi * stride;
        const float y = (float)yi;
        const float block_y = y / block_height;
        const float y_mod = y + my;
        const float weight_top = y_mod - truncf(y_mod);
        const float weight_bottom = 1.0f - weight_top;
        const uint32_t block_top = (uint32_t)floorf(block_y);
        const uint32_t block_bottom = truncf(y_mod) == 0.0f ? (uint32_t)ceilf(block_y) : block_top;
        for (uint32_t xi = 0; xi < width; xi++) {
            const uint8_t *p = row + (size_t)xi * (size_t)bpp;
            uint32_t v = (uint32_t)p[0] + p[1] + p[2];
            if (bpp == 4 && p[3] == 0)
                v = 765;
            const float px_sum = (float)v;
            const float x = (float)xi;
            const float block_x = x / block_width;
            const float x_mod = x + mx;
            const float weight_left = x_mod - truncf(x_mod);
            const float weight_right = 1.0f - weight_left;
            const uint32_t block_left = (uint32_t)floorf(block_x);
            const uint32_t block_right = truncf(x_mod) == 0.0f ? (uint32_t)ceilf(block_x) : block_left;
            if (block_left > 7 || block_right > 7 || block_top > 7 || block_bottom > 7)
                return ORC_ERR_PANIC; /* slice index out of bounds in the crate */
            blocks[block_top * 8 + block_left] += px_sum * weight_left * weight_top;
            blocks[block_bottom * 8 + block_left] += px_sum * weight_left * weight_bottom;
            blocks[block_top * 8 + block_right] += px_sum * weight_right * weight_top;
            blocks[block_bottom * 8 + block_right] += px_sum * weight_right * weight_bottom;
        }
    }
    memcpy(sums, blocks, sizeof(blocks));
    return ORC_OK;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

static int cmp_f32(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* gen_hash!: 4 groups of 16 blocks (2 block-rows); median = sorted[len/2] (upper median);
 * fast: bit = v > median || (v == median && median > 765*bw*bh/2)                       (u32)
 * slow: bit = v > median || (|v - median| < 0.001 && median > 765.0*(bw*bh)/2.0)         (f32, FLOAT_EQ_MARGIN) */
uint64_t orc_blockhash_bits(const uint32_t sums[64], uint32_t width, uint32_t height)
{
    uint64_t hash = 0;
    if (!blockhash_is_fast(width, height)) {
        float blocks[64];
        memcpy(blocks, sums, sizeof(blocks));
        const float block_width = (float)width / 8.0f, block_height = (float)height / 8.0f;
        const float block_area = block_width * block_height;
        const float cmp_factor = 765.0f * block_area / 2.0f;
        for (int band = 0; band < 4; band++) {
            float sorted[16];
            memcpy(sorted, blocks + band * 16, sizeof(sorted));
            qsort(sorted, 16, sizeof(float), cmp_f32);
            const float median = sorted[8];
            for (int i = 0; i < 16; i++) {
                const float v = blocks[band * 16 + i];
                if (v > median || (fabsf(v - median) < 0.001f && median > cmp_factor))
                    hash |= (uint64_t)1 << (band * 16 + i);
            }
        }
        return hash;
    }
    uint64_t half = (uint64_t)765 * (width / 8) * (height / 8) / 2;
    for (int band = 0; band < 4; band++) {
        uint32_t sorted[16];
        memcpy(sorted, sums + band * 16, sizeof(sorted));
        qsort(sorted, 16, sizeof(uint32_t), cmp_u32);
        uint32_t median = sorted[8];
        for (int i = 0; i < 16; i++) {
            uint32_t v = sums[band * 16 + i];
            int bit = v > median || (v == median && (uint64_t)median > half);
            if (bit)
                hash |= (uint64_t)1 << (band * 16 + i);
        }
    }
    return hash;
}

int orc_blockhash(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride,
                  int format, uint64_t *hash)
{
    uint32_t sums[64];
    int rc = orc_blockhash_sums(data, width, height, stride, format, sums);
    if (rc != ORC_OK)
        return rc;
    *hash = orc_blockhash_bits(sums, width, height);
    return ORC_OK;
}

uint32_t orc_hamming64(uint64_t a, uint64_t b) { return (uint32_t)__builtin_popcountll(a ^ b); }

/* ------------------------------------------------------------------ videocompare: Mean / Gradient /
 * VertGradient / DoubleGradient (HashAlg::{Mean,Gradient,VertGradient,DoubleGradient},
 * hashed_image.rs:89-107 -> image_hasher 3.1.1 on image 0.25.10; neither crate is under
 * /root/reference: PARITY UNPINNED, restated from the crates' published sources).
 *
 * image_hasher: hash_image = to_grayscale -> imageops::resize(gray, w', h', Lanczos3) -> bytes ->
 *   Mean (8x8):           mean = (sum / len) as u8; bit = px >= mean
 *   Gradient (9x8):       per row, bit = px[i] < px[i+1]
 *   VertGradient (8x9):   per column, bit = px[r] < px[r+1]
 *   DoubleGradient (5x5): Gradient bits followed by VertGradient bits (40 bits)
 * image::imageops::grayscale: Luma = (2126 r + 7152 g + 722 b) / 10000 in u32 (alpha ignored).
 * image::imageops::resize: same size => copy; else vertical_sample (u8 -> f32 rows) then
 *   horizontal_sample (f32 -> clamp -> round -> u8), weights lanczos3((i - centre) / sratio)
 *   normalised by their f32 running sum, accumulation `t += px * w` in f32, in tap order. */

static float lanczos_sinc(float t)
{
    float a = t * 3.14159274101257324f; /* f32::consts::PI */
    return t == 0.0f ? 1.0f : sinf(a) / a;
}

static float lanczos3_kernel(float x)
{
    return fabsf(x) < 3.0f ? lanczos_sinc(x) * lanczos_sinc(x / 3.0f) : 0.0f;
}

/* taps of output sample `out` when resampling `in_size` -> `out_size`: [*left, *left + n) and n normalised weights */
static uint32_t lanczos3_taps(uint32_t in_size, uint32_t out_size, uint32_t out, uint32_t *left_out, float *ws)
{
    float ratio = (float)in_size / (float)out_size;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = 3.0f * sratio;
    float input = ((float)out + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(input - src_support);
    if (left < 0) left = 0;
    if (left > (int64_t)in_size - 1) left = (int64_t)in_size - 1;
    int64_t right = (int64_t)ceilf(input + src_support);
    if (right < left + 1) right = left + 1;
    if (right > (int64_t)in_size) right = (int64_t)in_size;
    input = input - 0.5f;
    float sum = 0.0f;
    uint32_t n = 0;
    for (int64_t i = left; i < right; i++) {
        float w = lanczos3_kernel(((float)i - input) / sratio);
        ws[n++] = w;
        sum += w;
    }
    for (uint32_t k = 0; k < n; k++)
        ws[k] /= sum;
    *left_out = (uint32_t)left;
    return n;
}

int orc_gray_resize_lanczos3(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format,
                             uint32_t nw, uint32_t nh, uint8_t *out)
{
    int bpp;
    if (format == ORC_FORMAT_RGB) bpp = 3;
    else if (format == ORC_FORMAT_RGBA) bpp = 4;
    else return ORC_ERR_FORMAT;
    if (width == 0 || height == 0) { /* resize of an empty image: a zeroed buffer */
        memset(out, 0, (size_t)nw * nh);
        return ORC_OK;
    }
    uint8_t *gray = (uint8_t *)malloc((size_t)width * height);
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const uint8_t *p = data + (size_t)y * stride + (size_t)x * (size_t)bpp;
            uint32_t l = 2126u * p[0] + 7152u * p[1] + 722u * p[2];
            gray[(size_t)y * width + x] = (uint8_t)(l / 10000u);
        }
    if (nw == width && nh == height) { /* same dimensions: a copy, no resampling */
        memcpy(out, gray, (size_t)width * height);
        free(gray);
        return ORC_OK;
    }
    float *tmp = (float *)malloc((size_t)width * nh * sizeof(float));
    float *ws = (float *)malloc(((size_t)(height > width ? height : width) + 8) * sizeof(float));
    for (uint32_t oy = 0; oy < nh; oy++) { /* vertical_sample */
        uint32_t left;
        uint32_t n = lanczos3_taps(height, nh, oy, &left, ws);
        for (uint32_t x = 0; x < width; x++) {
            float t = 0.0f;
            for (uint32_t i = 0; i < n; i++)
                t += (float)gray[(size_t)(left + i) * width + x] * ws[i];
            tmp[(size_t)oy * width + x] = t;
        }
    }
    for (uint32_t ox = 0; ox < nw; ox++) { /* horizontal_sample */
        uint32_t left;
        uint32_t n = lanczos3_taps(width, nw, ox, &left, ws);
        for (uint32_t y = 0; y < nh; y++) {
            float t = 0.0f;
            for (uint32_t i = 0; i < n; i++)
                t += tmp[(size_t)y * width + left + i] * ws[i];
            t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t); /* NaN passes through; `as u8` makes it 0 */
            float r = roundf(t);
            out[(size_t)y * nw + ox] = r != r ? 0 : (uint8_t)r;
        }
    }
    free(ws);
    free(tmp);
    free(gray);
    return ORC_OK;
}

/* algo: 0 Mean, 1 Gradient, 2 VertGradient, 3 DoubleGradient (GstVideoCompareHashAlgorithm values,
 * videocompare/mod.rs:57-92).  bit k of *hash = k-th bool the crate's iterator yields. */
int orc_image_hash(const uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format, int algo,
                   uint64_t *hash, uint32_t *n_bits)
{
    uint32_t rw, rh;
    switch (algo) {
    case 0: rw = 8; rh = 8; break;
    case 1: rw = 9; rh = 8; break;
    case 2: rw = 8; rh = 9; break;
    case 3: rw = 5; rh = 5; break;
    default: return ORC_ERR_FORMAT;
    }
    uint8_t px[81];
    int rc = orc_gray_resize_lanczos3(data, width, height, stride, format, rw, rh, px);
    if (rc != ORC_OK) return rc;
    uint64_t h = 0;
    uint32_t k = 0;
    if (algo == 0) {
        uint32_t sum = 0;
        for (uint32_t i = 0; i < rw * rh; i++) sum += px[i];
        uint8_t mean = (uint8_t)(sum / (rw * rh));
        for (uint32_t i = 0; i < rw * rh; i++, k++)
            if (px[i] >= mean) h |= (uint64_t)1 << k;
    }
    if (algo == 1 || algo == 3)
        for (uint32_t y = 0; y < rh; y++)
            for (uint32_t x = 0; x + 1 < rw; x++, k++)
                if (px[y * rw + x] < px[y * rw + x + 1]) h |= (uint64_t)1 << k;
    if (algo == 2 || algo == 3)
        for (uint32_t x = 0; x < rw; x++)
            for (uint32_t y = 0; y + 1 < rh; y++, k++)
                if (px[y * rw + x] < px[(y + 1) * rw + x]) h |= (uint64_t)1 << k;
    *hash = h;
    *n_bits = k;
    return ORC_OK;
}

/* ------------------------------------------------------------------ imagersoverlay: gst_video_blend (libgstvideo 1.14.0)
 *
 * Behaviour of the library in the image, established by probing gst_video_overlay_composition_blend through ctypes on all
 * 65536 (source alpha, destination alpha) pairs, 1.5 M random colour samples, eight destination formats and seven global
 * alphas (0 mismatches; generator and fixtures: tests/golden/make_overlay_blend_golden.py):
 *   a_s  = overlay alpha;  with a rectangle global alpha g != 1:  a_s = a_s * (int)(g * 255) / 255      (integer division)
 *   a_s == 0  -> the destination pixel is left untouched
 *   a_d  = destination alpha byte; 255 for RGB / BGR.  The x byte of RGBx / BGRx / xRGB / xBGR IS treated as alpha by
 *          1.14.0 (read as a_d and overwritten with the result alpha) -- a quirk of that version, reproduced here.
 *   a_o  = a_s + a_d * (255 - a_s) / 255
 *   c_o  = (c_s * a_s + c_d * a_d * (255 - a_s) / 255) / max(a_o, 1)      per colour channel, all divisions truncating
 * The rectangle is clipped against the frame; overlay rows/columns that fall outside are skipped. */
int orc_overlay_blend(uint8_t *data, uint32_t width, uint32_t height, uint32_t stride, int format,
                      const uint8_t *overlay_bgra, uint32_t overlay_width, uint32_t overlay_height, uint32_t overlay_stride,
                      int32_t x, int32_t y, float global_alpha)
{
    int bpp, ir, ig, ib, ia; /* byte index of R, G, B and alpha (-1: none) in a destination pixel */
    switch (format) {
    case ORC_FORMAT_RGBA: case ORC_FORMAT_RGBX: bpp = 4; ir = 0; ig = 1; ib = 2; ia = 3; break;
    case ORC_FORMAT_BGRA: case ORC_FORMAT_BGRX: bpp = 4; ir = 2; ig = 1; ib = 0; ia = 3; break;
    case ORC_FORMAT_ARGB: case ORC_FORMAT_XRGB: bpp = 4; ir = 1; ig = 2; ib = 3; ia = 0; break;
    case ORC_FORMAT_ABGR: case ORC_FORMAT_XBGR: bpp = 4; ir = 3; ig = 2; ib = 1; ia = 0; break;
    case ORC_FORMAT_RGB: bpp = 3; ir = 0; ig = 1; ib = 2; ia = -1; break;
    case ORC_FORMAT_BGR: bpp = 3; ir = 2; ig = 1; ib = 0; ia = -1; break;
    default: return ORC_ERR_FORMAT;
    }
    const int have_g = global_alpha != 1.0f;
    const uint32_t g = (uint32_t)(int)(global_alpha * 255.0f);
    for (uint32_t oy = 0; oy < overlay_height; oy++) {
        const int64_t dy = (int64_t)y + oy;
        if (dy < 0 || dy >= (int64_t)height) continue;
        for (uint32_t ox = 0; ox < overlay_width; ox++) {
            const int64_t dx = (int64_t)x + ox;
            if (dx < 0 || dx >= (int64_t)width) continue;
            const uint8_t *s = overlay_bgra + (size_t)oy * overlay_stride + (size_t)ox * 4;
            uint8_t *d = data + (size_t)dy * stride + (size_t)dx * (size_t)bpp;
            uint32_t a_s = s[3];
            if (have_g) a_s = a_s * g / 255u;
            if (a_s == 0) continue;
            const uint32_t a_d = ia >= 0 ? d[ia] : 255u;
            const uint32_t a_o = a_s + a_d * (255u - a_s) / 255u;
            const uint32_t div = a_o ? a_o : 1u;
            const uint32_t cs[3] = {s[2], s[1], s[0]}; /* overlay is BGRA */
            const int idx[3] = {ir, ig, ib};
            for (int c = 0; c < 3; c++) {
                uint32_t v = (cs[c] * a_s + d[idx[c]] * a_d * (255u - a_s) / 255u) / div;
                d[idx[c]] = (uint8_t)(v > 255u ? 255u : v);
            }
            if (ia >= 0) d[ia] = (uint8_t)a_o;
        }
    }
    return ORC_OK;
}
