/*
 * oracle/colorlut_oracle.c -- CPU restatement of video/colorlut (TEST INFRASTRUCTURE ONLY).
 *
 * Follows /root/reference/video/colorlut/src/parser.rs (Adobe .cube parser) and
 * /root/reference/video/colorlut/src/colorlut/imp.rs:226-543 (1-D linear and 3-D trilinear
 * LUT on RGBA8 and RGBA64 LE/BE).  Build: gcc -O3 -ffp-contract=off.
 *
 * Rust semantics reproduced:
 *   str::lines()            -> split on '\n', a trailing '\r' is stripped
 *   str::trim / split_whitespace -> Unicode White_Space
 *   str::parse::<f32>       -> strict grammar (no hex, no trailing junk), correctly rounded
 *   str::parse::<usize>     -> optional '+', decimal digits only, overflow is an error
 *   f32::clamp (std)        -> NaN stays NaN            (imp.rs:471-479, 537-543)
 *   f32::round              -> half away from zero       (roundf)
 *   `as u8` / `as u16` / `as usize` -> saturating, NaN -> 0
 */
#include "oracle.h"

#include <errno.h>
#include <locale.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LUT_1D_MIN_SIZE 2u     /* parser.rs:12 */
#define LUT_1D_MAX_SIZE 65536u /* parser.rs:13 */
#define LUT_3D_MIN_SIZE 2u     /* parser.rs:15 */
#define LUT_3D_MAX_SIZE 256u   /* parser.rs:16 */

struct orc_cube_lut { /* parser.rs:68-74 CubeLut + :57-66 CubeLutKind */
    float domain_scale[3];
    float domain_offset[3];
    int is_3d;
    uint32_t size;
    float *rgba;     /* 3-D: size^3 * [r,g,b,1.0] (parser.rs:253-256) */
    float *table[3]; /* 1-D: r,g,b (parser.rs:226-236) */
};

/* ---- Unicode White_Space (char::is_whitespace) on UTF-8; returns byte length or 0 ---- */
static size_t ws_len(const unsigned char *p, const unsigned char *end)
{
    if (p >= end)
        return 0;
    unsigned c = p[0];
    if (c == ' ' || (c >= 0x09 && c <= 0x0d))
        return 1;
    if (c == 0xC2 && p + 1 < end && (p[1] == 0x85 || p[1] == 0xA0))
        return 2; /* U+0085, U+00A0 */
    if (c == 0xE1 && p + 2 < end && p[1] == 0x9A && p[2] == 0x80)
        return 3; /* U+1680 */
    if (c == 0xE2 && p + 2 < end) {
        if (p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 ||
                             p[2] == 0xAF))
            return 3; /* U+2000..U+200A, U+2028, U+2029, U+202F */
        if (p[1] == 0x81 && p[2] == 0x9F)
            return 3; /* U+205F */
    }
    if (c == 0xE3 && p + 2 < end && p[1] == 0x80 && p[2] == 0x80)
        return 3; /* U+3000 */
    return 0;
}

/* Rust's dec2flt grammar: [+-] ( inf | infinity | nan | digits [. digits] [e[+-]digits] ) */
static int rust_parse_f32(const char *s, size_t n, float *out)
{
    char buf[128];
    if (n == 0 || n >= sizeof(buf))
        return -1; /* tokens that long are never valid decimal floats in practice */
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-')
        i++;
    if (i == n)
        return -1;
    size_t rest = n - i;
    const char *q = s + i;
    int special = 0;
    if ((rest == 3 && strncasecmp(q, "inf", 3) == 0) ||
        (rest == 8 && strncasecmp(q, "infinity", 8) == 0) ||
        (rest == 3 && strncasecmp(q, "nan", 3) == 0))
        special = 1;
    if (!special) {
        size_t nd = 0;
        while (i < n && s[i] >= '0' && s[i] <= '9') { i++; nd++; }
        if (i < n && s[i] == '.') {
            i++;
            while (i < n && s[i] >= '0' && s[i] <= '9') { i++; nd++; }
        }
        if (nd == 0)
            return -1;
        if (i < n && (s[i] == 'e' || s[i] == 'E')) {
            i++;
            if (i < n && (s[i] == '+' || s[i] == '-'))
                i++;
            size_t ne = 0;
            while (i < n && s[i] >= '0' && s[i] <= '9') { i++; ne++; }
            if (ne == 0)
                return -1;
        }
        if (i != n)
            return -1;
    }
    memcpy(buf, s, n);
    buf[n] = 0;
    char *endp = NULL;
    /* glibc strtof is correctly rounded, like dec2flt; the _l form in the "C" locale keeps it independent of the
     * process LC_NUMERIC (Rust's str::parse is locale-free) */
    static locale_t c_locale;
    if (!c_locale) c_locale = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    float v = c_locale ? strtof_l(buf, &endp, c_locale) : strtof(buf, &endp);
    if (endp != buf + n)
        return -1;
    *out = v;
    return 0;
}

static int rust_parse_usize(const char *s, size_t n, uint64_t *out)
{
    size_t i = 0;
    if (n == 0)
        return -1;
    if (s[0] == '+')
        i = 1;
    if (i == n)
        return -1;
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9')
            return -1;
        uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10)
            return -1; /* overflow -> PosOverflow error */
        v = v * 10 + d;
    }
    *out = v;
    return 0;
}

typedef struct { const char *p; size_t n; } tok_t;

/* split_whitespace over [p,end): fills up to cap tokens, returns the total count */
static size_t split_ws(const unsigned char *p, const unsigned char *end, tok_t *toks, size_t cap)
{
    size_t count = 0;
    while (p < end) {
        size_t w = ws_len(p, end);
        if (w) { p += w; continue; }
        const unsigned char *start = p;
        while (p < end && !ws_len(p, end))
            p++;
        if (count < cap) {
            toks[count].p = (const char *)start;
            toks[count].n = (size_t)(p - start);
        }
        count++;
    }
    return count;
}

static void set_err(char *err, size_t cap, const char *fmt, size_t line_no)
{
    if (err && cap)
        snprintf(err, cap, fmt, line_no);
}

enum { ST_HEADER, ST_1D, ST_3D }; /* parser.rs:96-101 ParseState */

/* parser.rs:104-282 CubeLut::parse */
orc_cube_lut *orc_cube_parse(const char *text, size_t len, char *err, size_t cap)
{
    float domain_min[3] = {0.0f, 0.0f, 0.0f}; /* :112 */
    float domain_max[3] = {1.0f, 1.0f, 1.0f}; /* :113 */
    int state = ST_HEADER;
    int have_data = 0;
    uint64_t size = 0;
    float *values = NULL; /* Vec<[f32;3]> :116 */
    size_t n_values = 0, cap_values = 0;

    const unsigned char *p = (const unsigned char *)text;
    const unsigned char *end = p + len;
    size_t line_no = 0;

#define FAIL(msg) do { set_err(err, cap, msg, line_no); free(values); return NULL; } while (0)

    while (p < end) { /* text.lines() :118 */
        const unsigned char *nl = memchr(p, '\n', (size_t)(end - p));
        const unsigned char *line_end = nl ? nl : end;
        const unsigned char *next = nl ? nl + 1 : end;
        if (line_end > p && line_end[-1] == '\r' && nl)
            line_end--; /* lines() strips "\r\n" */
        line_no++;

        /* trim() :120 */
        const unsigned char *ls = p, *le = line_end;
        for (;;) {
            size_t w = ws_len(ls, le);
            if (!w) break;
            ls += w;
        }
        for (;;) { /* trailing: try 1..3 byte whitespace sequences */
            int trimmed = 0;
            for (size_t k = 1; k <= 3 && (size_t)(le - ls) >= k; k++) {
                if (ws_len(le - k, le) == k) { le -= k; trimmed = 1; break; }
            }
            if (!trimmed) break;
        }
        p = next;
        if (ls == le || *ls == '#') /* :121-123 */
            continue;

        tok_t toks[5];
        size_t nt = split_ws(ls, le, toks, 5); /* :125 */
        if (nt == 0)
            continue;
        const tok_t first = toks[0];
#define IS(kw) (first.n == sizeof(kw) - 1 && memcmp(first.p, kw, sizeof(kw) - 1) == 0)

        if (IS("TITLE")) { /* :133-135 */
            if (have_data) FAIL("Header found after LUT data at line %zu");
        } else if (IS("DOMAIN_MIN") || IS("DOMAIN_MAX")) { /* :136-143 */
            if (have_data) FAIL("Header found after LUT data at line %zu");
            float v[3]; /* parse_vec3 :317-334 */
            for (int c = 0; c < 3; c++) {
                if ((size_t)(c + 1) >= nt) FAIL("Invalid line %zu");
                if (rust_parse_f32(toks[c + 1].p, toks[c + 1].n, &v[c]) != 0)
                    FAIL("Invalid float at line %zu");
            }
            if (nt > 4) FAIL("Invalid line %zu");
            memcpy(IS("DOMAIN_MIN") ? domain_min : domain_max, v, sizeof(v));
        } else if (IS("LUT_1D_SIZE") || IS("LUT_3D_SIZE")) { /* :144-177 */
            int is1d = IS("LUT_1D_SIZE");
            if (have_data) FAIL("Header found after LUT data at line %zu");
            if (state != ST_HEADER) FAIL("Invalid LUT size keyword at line %zu");
            if (nt < 2) FAIL("Invalid line %zu"); /* parse_single_usize :336-357 */
            uint64_t v;
            if (rust_parse_usize(toks[1].p, toks[1].n, &v) != 0) FAIL("Invalid integer at line %zu");
            if (nt > 2) FAIL("Invalid line %zu");
            uint64_t lo = is1d ? LUT_1D_MIN_SIZE : LUT_3D_MIN_SIZE;
            uint64_t hi = is1d ? LUT_1D_MAX_SIZE : LUT_3D_MAX_SIZE;
            if (v < lo || v > hi) FAIL("Invalid LUT size at line %zu"); /* :303-315 */
            size = v;
            state = is1d ? ST_1D : ST_3D;
        } else { /* data row :178-201 */
            if (state == ST_HEADER) FAIL("LUT data found before LUT size at line %zu");
            have_data = 1;
            float v[3];
            for (int c = 0; c < 3; c++) {
                if ((size_t)c >= nt) FAIL("Invalid line %zu");
                if (rust_parse_f32(toks[c].p, toks[c].n, &v[c]) != 0)
                    FAIL("Invalid float at line %zu");
            }
            if (nt > 3) FAIL("Invalid line %zu"); /* :194-198 */
            if (n_values == cap_values) {
                cap_values = cap_values ? cap_values * 2 : 4096;
                float *nv = realloc(values, cap_values * 3 * sizeof(float));
                if (!nv) FAIL("out of memory at line %zu");
                values = nv;
            }
            memcpy(values + 3 * n_values, v, sizeof(v));
            n_values++;
        }
#undef IS
    }

    line_no = 0;
    /* :205-212; NaN compares false so a NaN domain passes, as in the reference */
    if (domain_min[0] >= domain_max[0] || domain_min[1] >= domain_max[1] ||
        domain_min[2] >= domain_max[2])
        FAIL("Invalid domain");
    if (state == ST_HEADER) FAIL("Missing LUT size"); /* :215-217 */

    orc_cube_lut *lut = calloc(1, sizeof(*lut));
    if (!lut) FAIL("out of memory");
    lut->size = (uint32_t)size;
    if (state == ST_1D) { /* :218-237 */
        if (n_values != size) { free(lut); FAIL("Invalid 1D LUT value count"); }
        lut->is_3d = 0;
        for (int c = 0; c < 3; c++) {
            lut->table[c] = malloc(size * sizeof(float));
            for (uint64_t i = 0; i < size; i++)
                lut->table[c][i] = values[3 * i + c];
        }
    } else { /* :238-261 */
        uint64_t expected = size * size * size;
        if (n_values != expected) { free(lut); FAIL("Invalid 3D LUT value count"); }
        lut->is_3d = 1;
        lut->rgba = malloc(expected * 4 * sizeof(float));
        for (uint64_t i = 0; i < expected; i++) {
            lut->rgba[4 * i + 0] = values[3 * i + 0];
            lut->rgba[4 * i + 1] = values[3 * i + 1];
            lut->rgba[4 * i + 2] = values[3 * i + 2];
            lut->rgba[4 * i + 3] = 1.0f; /* :253-256 */
        }
    }
    for (int c = 0; c < 3; c++) { /* :264-274 */
        lut->domain_scale[c] = 1.0f / (domain_max[c] - domain_min[c]);
        lut->domain_offset[c] = -domain_min[c] * lut->domain_scale[c];
    }
    free(values);
    return lut;
#undef FAIL
}

void orc_cube_free(orc_cube_lut *lut)
{
    if (!lut) return;
    free(lut->rgba);
    for (int c = 0; c < 3; c++) free(lut->table[c]);
    free(lut);
}

int orc_cube_is_3d(const orc_cube_lut *lut) { return lut->is_3d; }
uint32_t orc_cube_size(const orc_cube_lut *lut) { return lut->size; }
const float *orc_cube_domain_scale(const orc_cube_lut *lut) { return lut->domain_scale; }
const float *orc_cube_domain_offset(const orc_cube_lut *lut) { return lut->domain_offset; }
const float *orc_cube_rgba(const orc_cube_lut *lut) { return lut->rgba; }
const float *orc_cube_table_1d(const orc_cube_lut *lut, int c) { return lut->is_3d ? NULL : lut->table[c]; }

/* ---------------- colorlut/imp.rs per-pixel arithmetic ---------------- */

static inline float std_clamp01(float v) /* f32::clamp(0.0, 1.0): NaN propagates */
{
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}

static inline size_t f32_as_usize(float v) /* saturating, NaN -> 0 */
{
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)v;
}

static inline float norm_comp(const orc_cube_lut *lut, int c, uint8_t value) /* :471-474 */
{
    float v = (float)value / 255.0f;
    return std_clamp01(v * lut->domain_scale[c] + lut->domain_offset[c]);
}

static inline float norm_comp_u16(const orc_cube_lut *lut, int c, uint16_t value) /* :476-479 */
{
    float v = (float)value / 65535.0f;
    return std_clamp01(v * lut->domain_scale[c] + lut->domain_offset[c]);
}

static inline uint8_t float_to_u8(float v) /* :537-539 */
{
    float r = roundf(std_clamp01(v) * 255.0f);
    if (!(r == r) || r <= 0.0f) return 0;
    if (r >= 255.0f) return 255;
    return (uint8_t)r;
}

static inline uint16_t float_to_u16(float v) /* :541-543 */
{
    float r = roundf(std_clamp01(v) * 65535.0f);
    if (!(r == r) || r <= 0.0f) return 0;
    if (r >= 65535.0f) return 65535;
    return (uint16_t)r;
}

static inline float sample_1d(const float *lut, size_t len, float x) /* :482-490 */
{
    size_t max_idx = len - 1;
    size_t x0 = f32_as_usize(floorf(x));
    if (x0 > max_idx) x0 = max_idx;
    size_t x1 = x0 + 1;
    if (x1 > max_idx) x1 = max_idx;
    float t = x - (float)x0;
    return lut[x0] + (lut[x1] - lut[x0]) * t;
}

static inline void lerp4(const float a[4], const float b[4], float t, float out[4]) /* :528-535 */
{
    for (int i = 0; i < 4; i++)
        out[i] = a[i] + (b[i] - a[i]) * t;
}

static inline const float *at(const orc_cube_lut *lut, size_t x, size_t y, size_t z) /* parser.rs:43-53 */
{
    size_t s = lut->size;
    return lut->rgba + 4 * (x + y * s + z * s * s);
}

static void sample_3d(const orc_cube_lut *lut, float x, float y, float z, float out[4]) /* :493-526 */
{
    size_t max_idx = (size_t)lut->size - 1;
    size_t x0 = f32_as_usize(floorf(x)); if (x0 > max_idx) x0 = max_idx;
    size_t y0 = f32_as_usize(floorf(y)); if (y0 > max_idx) y0 = max_idx;
    size_t z0 = f32_as_usize(floorf(z)); if (z0 > max_idx) z0 = max_idx;
    size_t x1 = x0 + 1; if (x1 > max_idx) x1 = max_idx;
    size_t y1 = y0 + 1; if (y1 > max_idx) y1 = max_idx;
    size_t z1 = z0 + 1; if (z1 > max_idx) z1 = max_idx;
    float tx = x - (float)x0, ty = y - (float)y0, tz = z - (float)z0;

    float c00[4], c10[4], c01[4], c11[4], c0[4], c1[4];
    lerp4(at(lut, x0, y0, z0), at(lut, x1, y0, z0), tx, c00);
    lerp4(at(lut, x0, y1, z0), at(lut, x1, y1, z0), tx, c10);
    lerp4(at(lut, x0, y0, z1), at(lut, x1, y0, z1), tx, c01);
    lerp4(at(lut, x0, y1, z1), at(lut, x1, y1, z1), tx, c11);
    lerp4(c00, c10, ty, c0);
    lerp4(c01, c11, ty, c1);
    lerp4(c0, c1, tz, out);
}

static inline uint16_t rd16(const uint8_t *p, int le)
{
    return le ? (uint16_t)(p[0] | (p[1] << 8)) : (uint16_t)(p[1] | (p[0] << 8));
}

static inline void wr16(uint8_t *p, uint16_t v, int le)
{
    if (le) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
    else    { p[1] = (uint8_t)v; p[0] = (uint8_t)(v >> 8); }
}

/* imp.rs:203-223 dispatch + :237-397 row loops */
int orc_colorlut_transform_frame(const orc_cube_lut *lut, const uint8_t *src, size_t src_len,
                                 uint32_t src_stride, uint8_t *dst, size_t dst_len,
                                 uint32_t dst_stride, uint32_t width, uint32_t height,
                                 int format)
{
    if (!lut)
        return ORC_ERR_PARSE; /* "No LUT configured" :209-213 */
    int wide, le = 1;
    if (format == ORC_FORMAT_RGB10A2_LE) {
        /* RGB10A2_LE: the format d3d12colorlut accepts beside RGBA / RGBA64_LE (d3d12colorlut/imp.rs:236-244).  That element
         * samples the LUT in an HLSL shader (hardware filtering, not bit-defined); the CPU element's arithmetic is extended here
         * the way its 8- and 16-bit paths are written (imp.rs:471-479, 537-543): norm = v / 1023.0, same clamp / lattice /
         * trilinear steps, float_to_u10 = (clamp(v, 0, 1) * 1023.0).round() as u16, the 2 alpha bits copied.  Little-endian
         * dword: R bits 0-9, G 10-19, B 20-29, A 30-31.  Self-defined extension: PARITY UNPINNED by construction. */
        if (src_stride == 0 || dst_stride == 0) return ORC_ERR_PANIC;
        const float sm1 = (float)lut->size - 1.0f;
        for (size_t y = 0; y < height; y++) {
            if (y * (size_t)src_stride + (size_t)width * 4 > src_len || y * (size_t)dst_stride + (size_t)width * 4 > dst_len)
                return ORC_ERR_PANIC; /* &row[..width_in_bytes] out of range */
            const uint8_t *s = src + y * (size_t)src_stride;
            uint8_t *d = dst + y * (size_t)dst_stride;
            for (size_t x = 0; x < width; x++, s += 4, d += 4) {
                const uint32_t w = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
                float v[3], out[4];
                for (int c = 0; c < 3; c++) {
                    float n = (float)((w >> (10 * c)) & 1023u) / 1023.0f;
                    n = n * lut->domain_scale[c] + lut->domain_offset[c];
                    n = n < 0.0f ? 0.0f : n; /* f32::clamp: NaN stays NaN */
                    n = n > 1.0f ? 1.0f : n;
                    v[c] = n * sm1;
                }
                if (lut->is_3d) sample_3d(lut, v[0], v[1], v[2], out);
                else for (int c = 0; c < 3; c++) out[c] = sample_1d(lut->table[c], lut->size, v[c]);
                uint32_t o = w & 0xC0000000u;
                for (int c = 0; c < 3; c++) {
                    float f = out[c];
                    f = f < 0.0f ? 0.0f : f;
                    f = f > 1.0f ? 1.0f : f;
                    f = roundf(f * 1023.0f);
                    const uint32_t q = (f == f) ? (uint32_t)f : 0u; /* NaN as u16 == 0 */
                    o |= (q & 1023u) << (10 * c);
                }
                d[0] = (uint8_t)o; d[1] = (uint8_t)(o >> 8); d[2] = (uint8_t)(o >> 16); d[3] = (uint8_t)(o >> 24);
            }
        }
        return ORC_OK;
    }
    switch (format) {
    case ORC_FORMAT_RGBA: wide = 0; break;
    case ORC_FORMAT_RGBA64_LE: wide = 1; le = 1; break;
    case ORC_FORMAT_RGBA64_BE: wide = 1; le = 0; break;
    default: return ORC_ERR_FORMAT; /* unreachable!() :219 */
    }
    if (src_stride == 0 || dst_stride == 0)
        return ORC_ERR_PANIC;
    const size_t bpp = wide ? 8 : 4;
    const size_t row_bytes = (size_t)width * bpp;
    const float sm1 = (float)lut->size - 1.0f; /* `size as f32 - 1.0` :408,:438 */

    /* chunks(stride).take(height), zipped */
    size_t src_rows = (src_len + src_stride - 1) / src_stride;
    size_t dst_rows = (dst_len + dst_stride - 1) / dst_stride;
    size_t rows = height;
    if (src_rows < rows) rows = src_rows;
    if (dst_rows < rows) rows = dst_rows;
    for (size_t y = 0; y < rows; y++) {
        size_t so = y * (size_t)src_stride, dofs = y * (size_t)dst_stride;
        size_t s_avail = src_len - so < src_stride ? src_len - so : src_stride;
        size_t d_avail = dst_len - dofs < dst_stride ? dst_len - dofs : dst_stride;
        if (row_bytes > s_avail || row_bytes > d_avail)
            return ORC_ERR_PANIC; /* &row[..width_in_bytes] out of range */
        const uint8_t *s = src + so;
        uint8_t *d = dst + dofs;
        for (size_t x = 0; x < width; x++, s += bpp, d += bpp) {
            if (!wide) {
                if (!lut->is_3d) { /* transform_rgba_1d :237-265, apply_1d :399-413 */
                    for (int c = 0; c < 3; c++) {
                        float xx = norm_comp(lut, c, s[c]) * sm1;
                        d[c] = float_to_u8(sample_1d(lut->table[c], lut->size, xx));
                    }
                } else { /* transform_rgba_3d :267-294, apply_3d :431-449 */
                    float out[4];
                    float xx = norm_comp(lut, 0, s[0]) * sm1;
                    float yy = norm_comp(lut, 1, s[1]) * sm1;
                    float zz = norm_comp(lut, 2, s[2]) * sm1;
                    sample_3d(lut, xx, yy, zz, out);
                    d[0] = float_to_u8(out[0]);
                    d[1] = float_to_u8(out[1]);
                    d[2] = float_to_u8(out[2]);
                }
                d[3] = s[3]; /* :262,:291 */
            } else {
                if (!lut->is_3d) { /* transform_rgba64_1d :308-349, apply_1d_u16 :415-429 */
                    for (int c = 0; c < 3; c++) {
                        uint16_t v = rd16(s + 2 * c, le);
                        float xx = norm_comp_u16(lut, c, v) * sm1;
                        wr16(d + 2 * c, float_to_u16(sample_1d(lut->table[c], lut->size, xx)), le);
                    }
                } else { /* transform_rgba64_3d :351-397, apply_3d_u16 :451-469 */
                    float out[4];
                    float xx = norm_comp_u16(lut, 0, rd16(s + 0, le)) * sm1;
                    float yy = norm_comp_u16(lut, 1, rd16(s + 2, le)) * sm1;
                    float zz = norm_comp_u16(lut, 2, rd16(s + 4, le)) * sm1;
                    sample_3d(lut, xx, yy, zz, out);
                    wr16(d + 0, float_to_u16(out[0]), le);
                    wr16(d + 2, float_to_u16(out[1]), le);
                    wr16(d + 4, float_to_u16(out[2]), le);
                }
                d[6] = s[6]; /* alpha word copied raw :346,:394 */
                d[7] = s[7];
            }
        }
    }
    return ORC_OK;
}
