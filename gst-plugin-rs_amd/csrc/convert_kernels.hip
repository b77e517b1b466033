// I420 <-> RGBA converters for gfx950: what `videoconvert` does on either side of the filters in the reference's own
// example pipelines (video/colorlut/src/colorlut/imp.rs:17-19), bit-exact with GStreamer 1.14.0's
// default-caps behaviour (arithmetic and its provenance: convert_math.hpp; oracle: oracle/convert_oracle.c, pinned
// against the real element).
//
// Layout: one lane owns an 8 x 2 pixel tile = 4 chroma samples: 2 x 8 luma bytes (two 8-byte accesses), 4 + 4 chroma
// bytes (two dword accesses) and 2 x 32 RGBA bytes (four 16-byte accesses), so a wave covers 512 contiguous pixels of
// two rows.  HBM traffic 1.5 + 4 = 5.5 B/px either way; no reuse across lanes except the co-sited chroma filter's
// left neighbour column (one extra dword per row and lane, an L1/L2 hit).  Frames that are not aligned for these
// accesses, and the last partial tile of a row, take the per-sample path of the same kernel.
#include "mvfx_internal.h"

#include "convert_math.hpp"

#include <algorithm>

namespace mvfx {
namespace {

constexpr int kCvtBlock = 256;

// one launch covers one frame from each of up to kMaxBatch streams (blockIdx.z): a 4K frame is ~8 us of traffic, so
// single-frame launches are bound by the launch itself (13.4 us)
struct PlanesIn {
    const uint8_t *y[kMaxBatch], *u[kMaxBatch], *v[kMaxBatch];
    uint64_t ys, us, vs;
};
struct PlanesOut {
    uint8_t *y[kMaxBatch], *u[kMaxBatch], *v[kMaxBatch];
    uint64_t ys, us, vs;
};
struct PackedBatch {
    uint8_t *base[kMaxBatch];
};

// COLS columns x 2 rows per lane.  8: 8-byte luma loads, but every 16-byte RGBA store instruction leaves 16-byte holes between the
// lanes (a lane owns 32 bytes of a row); 4: 4-byte luma / 2-byte chroma loads and one fully contiguous 16-byte store per row.
template <bool ALIGNED, int COLS>
__global__ __launch_bounds__(kCvtBlock) void i420_to_rgba_kernel(PlanesIn in, uint32_t width, uint32_t height, YuvToRgbCoef k,
                                                                 PackedBatch outs, uint64_t out_stride)
{
    uint8_t *out = outs.base[blockIdx.z];
    const uint32_t x0 = (blockIdx.x * kCvtBlock + threadIdx.x) * COLS;
    const uint32_t y0 = blockIdx.y * 2;
    if (x0 >= width) return;
    const bool row1 = y0 + 1 < height;
    const uint8_t *yr0 = in.y[blockIdx.z] + (uint64_t)y0 * in.ys, *yr1 = yr0 + in.ys;
    const uint8_t *ur = in.u[blockIdx.z] + (uint64_t)blockIdx.y * in.us, *vr = in.v[blockIdx.z] + (uint64_t)blockIdx.y * in.vs;
    uint8_t *o0 = out + (uint64_t)y0 * out_stride, *o1 = o0 + out_stride;
    if (ALIGNED && x0 + COLS <= width) {
        uint32_t ya[2] = {0, 0}, yb[2] = {0, 0}, u4, v4;
        if constexpr (COLS == 8) {
            const uint2 a = *reinterpret_cast<const uint2 *>(yr0 + x0);
            const uint2 b = row1 ? *reinterpret_cast<const uint2 *>(yr1 + x0) : make_uint2(0, 0);
            ya[0] = a.x; ya[1] = a.y; yb[0] = b.x; yb[1] = b.y;
            u4 = *reinterpret_cast<const uint32_t *>(ur + x0 / 2);
            v4 = *reinterpret_cast<const uint32_t *>(vr + x0 / 2);
        } else {
            ya[0] = *reinterpret_cast<const uint32_t *>(yr0 + x0);
            yb[0] = row1 ? *reinterpret_cast<const uint32_t *>(yr1 + x0) : 0u;
            u4 = *reinterpret_cast<const uint16_t *>(ur + x0 / 2);
            v4 = *reinterpret_cast<const uint16_t *>(vr + x0 / 2);
        }
        uint32_t pa[COLS], pb[COLS];
#pragma unroll
        for (int j = 0; j < COLS / 2; j++) {
            const ChromaTerms c = chroma_terms((u4 >> (8 * j)) & 0xffu, (v4 >> (8 * j)) & 0xffu, k);
            const uint32_t wa = ya[j / 2], wb = yb[j / 2];
            const int s = (2 * j & 3) * 8;
            pa[2 * j] = yuv_pixel((wa >> s) & 0xffu, c, k);
            pa[2 * j + 1] = yuv_pixel((wa >> (s + 8)) & 0xffu, c, k);
            pb[2 * j] = yuv_pixel((wb >> s) & 0xffu, c, k);
            pb[2 * j + 1] = yuv_pixel((wb >> (s + 8)) & 0xffu, c, k);
        }
        uint4 *d0 = reinterpret_cast<uint4 *>(o0 + (uint64_t)x0 * 4);
#pragma unroll
        for (int q = 0; q < COLS / 4; q++) d0[q] = make_uint4(pa[4 * q], pa[4 * q + 1], pa[4 * q + 2], pa[4 * q + 3]);
        if (row1) {
            uint4 *d1 = reinterpret_cast<uint4 *>(o1 + (uint64_t)x0 * 4);
#pragma unroll
            for (int q = 0; q < COLS / 4; q++) d1[q] = make_uint4(pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]);
        }
        return;
    }
    for (uint32_t x = x0; x < min(x0 + COLS, width); x++) {
        const ChromaTerms c = chroma_terms(ur[x / 2], vr[x / 2], k);
        const uint32_t a = yuv_pixel(yr0[x], c, k);
        uint8_t *q = o0 + (uint64_t)x * 4;
        q[0] = (uint8_t)a; q[1] = (uint8_t)(a >> 8); q[2] = (uint8_t)(a >> 16); q[3] = 255;
        if (row1) {
            const uint32_t b = yuv_pixel(yr1[x], c, k);
            q = o1 + (uint64_t)x * 4;
            q[0] = (uint8_t)b; q[1] = (uint8_t)(b >> 8); q[2] = (uint8_t)(b >> 16); q[3] = 255;
        }
    }
}

__device__ __forceinline__ uint32_t load_px(const uint8_t *p, bool dword)
{
    if (dword) return *reinterpret_cast<const uint32_t *>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// horizontal chroma filter of video-chroma.c for sample ci of cw: l / c / r = the vertically averaged values of the
// columns 2ci-1, 2ci, 2ci+1
__device__ __forceinline__ uint32_t chroma_h(int32_t l, int32_t c, int32_t r, uint32_t ci, uint32_t cw, bool cosited)
{
    if (!cosited) return (uint32_t)((c + r + 1) >> 1);
    if (ci == 0) return (uint32_t)((3 * c + r + 2) >> 2);
    if (ci == cw - 1) return (uint32_t)((l + 3 * c + 2) >> 2);
    return (uint32_t)((l + 2 * c + r + 2) >> 2);
}

template <bool ALIGNED, int COLS>
__global__ __launch_bounds__(kCvtBlock) void rgba_to_i420_kernel(PackedBatch ins, uint64_t in_stride, uint32_t width, uint32_t height,
                                                                 RgbToYuvCoef k, PlanesOut out, bool dword_ok)
{
    const uint8_t *in = ins.base[blockIdx.z];
    const uint32_t x0 = (blockIdx.x * kCvtBlock + threadIdx.x) * COLS;
    const uint32_t y0 = blockIdx.y * 2;
    if (x0 >= width) return;
    const uint8_t *r0 = in + (uint64_t)y0 * in_stride, *r1 = r0 + in_stride;
    uint8_t *oy0 = out.y[blockIdx.z] + (uint64_t)y0 * out.ys, *oy1 = oy0 + out.ys;
    uint8_t *ou = out.u[blockIdx.z] + (uint64_t)blockIdx.y * out.us, *ov = out.v[blockIdx.z] + (uint64_t)blockIdx.y * out.vs;
    const uint32_t cw = width / 2;
    const bool cosited = k.cosited != 0;
    if (ALIGNED && x0 + COLS <= width) {
        uint32_t pa[COLS], pb[COLS];
#pragma unroll
        for (int q = 0; q < COLS / 4; q++) { // COLS = 4: one contiguous 16-byte load per row and lane
            const uint4 a = reinterpret_cast<const uint4 *>(r0 + (uint64_t)x0 * 4)[q], b = reinterpret_cast<const uint4 *>(r1 + (uint64_t)x0 * 4)[q];
            pa[4 * q] = a.x; pa[4 * q + 1] = a.y; pa[4 * q + 2] = a.z; pa[4 * q + 3] = a.w;
            pb[4 * q] = b.x; pb[4 * q + 1] = b.y; pb[4 * q + 2] = b.z; pb[4 * q + 3] = b.w;
        }
        uint32_t ya[COLS / 4] = {}, yb[COLS / 4] = {};
        int32_t cu[COLS], cv[COLS];
#pragma unroll
        for (int j = 0; j < COLS; j++) {
            const uint32_t la = rgb_luma(pa[j], k), lb = rgb_luma(pb[j], k);
            ya[j / 4] |= la << (8 * (j & 3));
            yb[j / 4] |= lb << (8 * (j & 3));
            cu[j] = (rgb_u(pa[j], k) + rgb_u(pb[j], k) + 1) >> 1; // vertical first
            cv[j] = (rgb_v(pa[j], k) + rgb_v(pb[j], k) + 1) >> 1;
        }
        if constexpr (COLS == 8) {
            *reinterpret_cast<uint2 *>(oy0 + x0) = make_uint2(ya[0], ya[1]);
            *reinterpret_cast<uint2 *>(oy1 + x0) = make_uint2(yb[0], yb[1]);
        } else {
            *reinterpret_cast<uint32_t *>(oy0 + x0) = ya[0];
            *reinterpret_cast<uint32_t *>(oy1 + x0) = yb[0];
        }
        int32_t lu = 0, lv = 0; // column x0 - 1 (co-sited filter only)
        if (cosited && x0 > 0) {
            const uint32_t qa = *reinterpret_cast<const uint32_t *>(r0 + (uint64_t)(x0 - 1) * 4), qb = *reinterpret_cast<const uint32_t *>(r1 + (uint64_t)(x0 - 1) * 4);
            lu = (rgb_u(qa, k) + rgb_u(qb, k) + 1) >> 1;
            lv = (rgb_v(qa, k) + rgb_v(qb, k) + 1) >> 1;
        }
        uint32_t u4 = 0, v4 = 0;
#pragma unroll
        for (int i = 0; i < COLS / 2; i++) {
            const uint32_t ci = x0 / 2 + i;
            u4 |= chroma_h(i ? cu[2 * i - 1] : lu, cu[2 * i], cu[2 * i + 1], ci, cw, cosited) << (8 * i);
            v4 |= chroma_h(i ? cv[2 * i - 1] : lv, cv[2 * i], cv[2 * i + 1], ci, cw, cosited) << (8 * i);
        }
        if constexpr (COLS == 8) {
            *reinterpret_cast<uint32_t *>(ou + x0 / 2) = u4;
            *reinterpret_cast<uint32_t *>(ov + x0 / 2) = v4;
        } else {
            *reinterpret_cast<uint16_t *>(ou + x0 / 2) = (uint16_t)u4;
            *reinterpret_cast<uint16_t *>(ov + x0 / 2) = (uint16_t)v4;
        }
        return;
    }
    // per-sample path: 2 x 2 block per chroma sample, neighbours read again
    for (uint32_t x = x0; x < min(x0 + COLS, width); x += 2) {
        const uint32_t ci = x / 2;
        int32_t vu[3] = {0, 0, 0}, vv[3] = {0, 0, 0}; // columns x-1, x, x+1
        for (int d = -1; d <= 1; d++) {
            if ((d < 0 && x == 0)) continue;
            const uint32_t xx = x + d; // x + 1 < width because width is even
            const uint32_t qa = load_px(r0 + (uint64_t)xx * 4, dword_ok), qb = load_px(r1 + (uint64_t)xx * 4, dword_ok);
            vu[d + 1] = (rgb_u(qa, k) + rgb_u(qb, k) + 1) >> 1;
            vv[d + 1] = (rgb_v(qa, k) + rgb_v(qb, k) + 1) >> 1;
            if (d >= 0) {
                oy0[xx] = (uint8_t)rgb_luma(qa, k);
                oy1[xx] = (uint8_t)rgb_luma(qb, k);
            }
        }
        ou[ci] = (uint8_t)chroma_h(vu[0], vu[1], vu[2], ci, cw, cosited);
        ov[ci] = (uint8_t)chroma_h(vv[0], vv[1], vv[2], ci, cw, cosited);
    }
}

// Any frame size and both 4:2:0 layouts (round 3): one lane per chroma sample.  GStreamer's videoconvert treats an odd-sized frame as
// the next even size with the last column / row replicated (pinned against the element at 65x33, 3x3, 5x7, 7x601, 641x481, 1919x1079
// ...: tests/golden/videoconvert_kat.npz); NV12 carries the same U and V as one plane of interleaved pairs.  Used for odd sizes and
// for NV12 output; even-sized I420 keeps the 8-columns-per-lane kernel above.
template <bool NV12>
__global__ __launch_bounds__(kCvtBlock) void rgba_to_yuv420_any_kernel(PackedBatch ins, uint64_t in_stride, uint32_t width, uint32_t height,
                                                                       RgbToYuvCoef k, PlanesOut out, bool dword_ok)
{
    const uint8_t *in = ins.base[blockIdx.z];
    const uint32_t cw = (width + 1) / 2;
    const uint32_t ci = blockIdx.x * kCvtBlock + threadIdx.x, cj = blockIdx.y;
    if (ci >= cw) return;
    const bool cosited = k.cosited != 0;
    const uint32_t y0 = 2 * cj, y1 = min(y0 + 1, height - 1);
    const uint8_t *r0 = in + (uint64_t)y0 * in_stride, *r1 = in + (uint64_t)y1 * in_stride;
    int32_t vu[3], vv[3]; // the vertically averaged columns 2ci-1, 2ci, 2ci+1 of the replicated frame
#pragma unroll
    for (int d = -1; d <= 1; d++) {
        const uint32_t xx = (uint32_t)min(max((int32_t)(2 * ci) + d, 0), (int32_t)width - 1);
        const uint32_t qa = load_px(r0 + (uint64_t)xx * 4, dword_ok), qb = load_px(r1 + (uint64_t)xx * 4, dword_ok);
        vu[d + 1] = (rgb_u(qa, k) + rgb_u(qb, k) + 1) >> 1; // vertical first
        vv[d + 1] = (rgb_v(qa, k) + rgb_v(qb, k) + 1) >> 1;
        if (d >= 0 && 2 * ci + d < width) { // this lane's luma samples (the replicated column / row is not part of the picture)
            out.y[blockIdx.z][(uint64_t)y0 * out.ys + xx] = (uint8_t)rgb_luma(qa, k);
            if (y0 + 1 < height) out.y[blockIdx.z][(uint64_t)(y0 + 1) * out.ys + xx] = (uint8_t)rgb_luma(qb, k);
        }
    }
    const uint32_t u = chroma_h(vu[0], vu[1], vu[2], ci, cw, cosited), v = chroma_h(vv[0], vv[1], vv[2], ci, cw, cosited);
    if (NV12) {
        uint8_t *o = out.u[blockIdx.z] + (uint64_t)cj * out.us + 2 * (uint64_t)ci;
        o[0] = (uint8_t)u;
        o[1] = (uint8_t)v;
    } else {
        out.u[blockIdx.z][(uint64_t)cj * out.us + ci] = (uint8_t)u;
        out.v[blockIdx.z][(uint64_t)cj * out.vs + ci] = (uint8_t)v;
    }
}

// NV12 -> RGBA as GStreamer 1.14.0's videoconvert does it (round 3; oracle/convert_oracle.c orc_convert_nv12_to_rgba, pinned against
// the element): there is no fast path for NV12, the chroma is INTERPOLATED -- horizontally first, then vertically -- before the same
// saturating 16-bit matrix as I420 -> RGBA.  One lane = one pixel row x 4 columns (one 16-byte store); the lane reads the (U, V) pairs
// k-1 .. k+2 of its chroma row and of the neighbouring chroma row the vertical filter pairs it with.
__device__ __forceinline__ uint32_t nv12_h(uint32_t prev, uint32_t cur, uint32_t next, uint32_t x, uint32_t w, bool cosited)
{
    // per byte lane of a packed (U | V << 16) value: pixel pairs (i, i+1), i odd, i < w-1 -> (3a + b + 2) >> 2, (a + 3b + 2) >> 2;
    // co-sited: odd pixels i < w-1 -> (a + b + 1) >> 1.  U and V sit 16 bits apart: the sums (<= 1022) never carry across.
    if (cosited) return ((x & 1) && x < w - 1) ? ((cur + next + 0x00010001u) >> 1) & 0x00ff00ffu : cur;
    if ((x & 1) && x < w - 1) return ((3 * cur + next + 0x00020002u) >> 2) & 0x00ff00ffu;
    if (!(x & 1) && x >= 2) return ((prev + 3 * cur + 0x00020002u) >> 2) & 0x00ff00ffu;
    return cur;
}

__global__ __launch_bounds__(kCvtBlock) void nv12_to_rgba_kernel(const uint8_t *yp, const uint8_t *uvp, uint64_t ys, uint64_t uvs, uint32_t width,
                                                                 uint32_t height, YuvToRgbCoef k, bool cosited, uint8_t *out, uint64_t out_stride,
                                                                 bool aligned)
{
    const uint32_t x0 = (blockIdx.x * kCvtBlock + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= width) return;
    const uint32_t cw = (width + 1) / 2, cy = y / 2;
    // the chroma row the vertical filter combines with this one: row pairs (j, j+1), j odd, j < h-1
    const bool v_next = (y & 1) && y < height - 1, v_prev = !(y & 1) && y >= 2;
    const uint8_t *ra = uvp + (uint64_t)cy * uvs, *rb = v_next ? ra + uvs : (v_prev ? ra - uvs : ra);
    uint32_t ca[4], cb[4]; // (U | V << 16) of the chroma samples x0/2 - 1 .. x0/2 + 2, clamped to the row
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t kx = (uint32_t)min(max((int32_t)(x0 / 2) - 1 + i, 0), (int32_t)cw - 1);
        const uint32_t a = (uint32_t)ra[2 * (uint64_t)kx] | ((uint32_t)ra[2 * (uint64_t)kx + 1] << 16);
        const uint32_t b = (uint32_t)rb[2 * (uint64_t)kx] | ((uint32_t)rb[2 * (uint64_t)kx + 1] << 16);
        ca[i] = a;
        cb[i] = b;
    }
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t x = x0 + j;
        if (x >= width) { px[j] = 0; continue; }
        const int i = 1 + j / 2; // this pixel's own chroma sample within ca / cb
        const uint32_t ha = nv12_h(ca[i - 1], ca[i], ca[i + 1 < 4 ? i + 1 : 3], x, width, cosited);
        uint32_t c = ha;
        if (v_next || v_prev) {
            const uint32_t hb = nv12_h(cb[i - 1], cb[i], cb[i + 1 < 4 ? i + 1 : 3], x, width, cosited);
            c = ((3 * ha + hb + 0x00020002u) >> 2) & 0x00ff00ffu; // own row weighs 3 in both halves of a pair
        }
        const ChromaTerms t = chroma_terms(c & 0xffu, c >> 16, k);
        px[j] = yuv_pixel(yp[(uint64_t)y * ys + x], t, k);
    }
    uint8_t *o = out + (uint64_t)y * out_stride + (uint64_t)x0 * 4;
    if (aligned && x0 + 4 <= width) {
        *reinterpret_cast<uint4 *>(o) = make_uint4(px[0], px[1], px[2], px[3]);
    } else {
        for (int j = 0; j < 4 && x0 + j < width; j++) {
            o[4 * j] = (uint8_t)px[j]; o[4 * j + 1] = (uint8_t)(px[j] >> 8); o[4 * j + 2] = (uint8_t)(px[j] >> 16); o[4 * j + 3] = 255;
        }
    }
}

int check_nv12(const mvfx_planar_frame *f, const char *what)
{
    if (!f)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL frame", what);
    if (f->format != MVFX_FORMAT_NV12)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "%s: planar format %d is not NV12", what, f->format);
    if (f->width == 0 || f->height == 0)
        return MVFX_OK;
    if (!f->data[0] || !f->data[1])
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL plane", what);
    if (f->stride[0] < f->width || f->stride[1] < 2 * ((f->width + 1) / 2))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: plane stride smaller than its row", what);
    return MVFX_OK;
}

int check_i420(const mvfx_planar_frame *f, const char *what)
{
    if (!f)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL frame", what);
    if (f->format != MVFX_FORMAT_I420)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "%s: planar format %d is not I420", what, f->format);
    if (f->width == 0 || f->height == 0)
        return MVFX_OK;
    const uint32_t cw = (f->width + 1) / 2;
    if (!f->data[0] || !f->data[1] || !f->data[2])
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL plane", what);
    if (f->stride[0] < f->width || f->stride[1] < cw || f->stride[2] < cw)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: plane stride smaller than its row", what);
    return MVFX_OK;
}

} // namespace
} // namespace mvfx

using namespace mvfx;

// n frame pairs sharing geometry, strides and formats (n == 1: one frame); <= kMaxBatch per launch
static int i420_to_rgba_impl(const mvfx_planar_frame *ins, const mvfx_frame *outs, uint32_t n, int32_t yuv_standard, hipStream_t st)
{
    if (!ins || !outs || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: NULL frame or empty batch");
    const mvfx_planar_frame *i420_in = &ins[0];
    const mvfx_frame *rgba_out = &outs[0];
    for (uint32_t i = 0; i < n; i++) {
        if (int rc = check_i420(&ins[i], "convert input"); rc != MVFX_OK) return rc;
        if (int rc = check_packed_frame(&outs[i], "convert output"); rc != MVFX_OK) return rc;
        if (outs[i].format != MVFX_FORMAT_RGBA)
            return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "convert: output format %d is not RGBA", outs[i].format);
        if (ins[i].width != outs[i].width || ins[i].height != outs[i].height)
            return fail(MVFX_ERR_NOT_NEGOTIATED, "convert: input %ux%u and output %ux%u differ", ins[i].width, ins[i].height, outs[i].width, outs[i].height);
        if (ins[i].width != i420_in->width || ins[i].height != i420_in->height || outs[i].stride != rgba_out->stride ||
            ins[i].stride[0] != i420_in->stride[0] || ins[i].stride[1] != i420_in->stride[1] || ins[i].stride[2] != i420_in->stride[2])
            return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: frames of one batch must share geometry and strides");
    }
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: yuv_standard %d is not 0 (by height), 1 (BT.601), 2 (BT.709), 3 (BT.2020)", yuv_standard);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t w = i420_in->width, h = i420_in->height;
    if (w == 0 || h == 0) return MVFX_OK;
    const YuvToRgbCoef k = yuv_to_rgb_coef(pick_yuv_standard(h, yuv_standard));
    const uint32_t rows2 = (h + 1) / 2;
    if (rows2 > 65535u)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: height %u too large", h);
    for (uint32_t done = 0; done < n; done += kMaxBatch) {
        const uint32_t m = std::min<uint32_t>(kMaxBatch, n - done);
        PlanesIn in{};
        PackedBatch out{};
        in.ys = i420_in->stride[0]; in.us = i420_in->stride[1]; in.vs = i420_in->stride[2];
        uint64_t a8 = in.ys, a4 = in.us | in.vs, a16 = rgba_out->stride;
        for (uint32_t i = 0; i < m; i++) {
            in.y[i] = static_cast<const uint8_t *>(ins[done + i].data[0]);
            in.u[i] = static_cast<const uint8_t *>(ins[done + i].data[1]);
            in.v[i] = static_cast<const uint8_t *>(ins[done + i].data[2]);
            out.base[i] = static_cast<uint8_t *>(outs[done + i].data);
            a8 |= reinterpret_cast<uintptr_t>(in.y[i]);
            a4 |= reinterpret_cast<uintptr_t>(in.u[i]) | reinterpret_cast<uintptr_t>(in.v[i]);
            a16 |= reinterpret_cast<uintptr_t>(out.base[i]);
        }
        const bool aligned = (a8 & 7) == 0 && (a4 & 3) == 0 && (a16 & 15) == 0;
        // four columns per lane: 16 frames per launch 108.7 k -> 120.5 k fps (0.62 -> 0.69 of HBM peak; another box 0.66 -> 0.74), one
        // frame 13.9 -> 13.0 us -- the write side is 73 % of this kernel's traffic and wants contiguous store instructions
        constexpr int kCols = 4;
        const dim3 grid(((w + kCols - 1) / kCols + kCvtBlock - 1) / kCvtBlock, rows2, m);
        if (aligned)
            MVFX_LAUNCH((i420_to_rgba_kernel<true, kCols>), grid, dim3(kCvtBlock), 0, st, in, w, h, k, out, (uint64_t)rgba_out->stride);
        else
            MVFX_LAUNCH((i420_to_rgba_kernel<false, kCols>), grid, dim3(kCvtBlock), 0, st, in, w, h, k, out, (uint64_t)rgba_out->stride);
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}

static int rgba_to_i420_impl(const mvfx_frame *ins, const mvfx_planar_frame *outs, uint32_t n, int32_t yuv_standard, hipStream_t st,
                             bool nv12 = false)
{
    if (!ins || !outs || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: NULL frame or empty batch");
    const mvfx_frame *rgba_in = &ins[0];
    const mvfx_planar_frame *i420_out = &outs[0];
    for (uint32_t i = 0; i < n; i++) {
        if (int rc = check_packed_frame(&ins[i], "convert input"); rc != MVFX_OK) return rc;
        if (int rc = nv12 ? check_nv12(&outs[i], "convert output") : check_i420(&outs[i], "convert output"); rc != MVFX_OK) return rc;
        if (ins[i].format != MVFX_FORMAT_RGBA)
            return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "convert: input format %d is not RGBA", ins[i].format);
        if (outs[i].width != ins[i].width || outs[i].height != ins[i].height)
            return fail(MVFX_ERR_NOT_NEGOTIATED, "convert: input %ux%u and output %ux%u differ", ins[i].width, ins[i].height, outs[i].width, outs[i].height);
        if (ins[i].width != rgba_in->width || ins[i].height != rgba_in->height || ins[i].stride != rgba_in->stride ||
            outs[i].stride[0] != i420_out->stride[0] || outs[i].stride[1] != i420_out->stride[1] ||
            (!nv12 && outs[i].stride[2] != i420_out->stride[2]))
            return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: frames of one batch must share geometry and strides");
    }
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: yuv_standard %d is not 0 (by height), 1 (BT.601), 2 (BT.709), 3 (BT.2020)", yuv_standard);
    const uint32_t w = rgba_in->width, h = rgba_in->height;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (w == 0 || h == 0) return MVFX_OK;
    const RgbToYuvCoef k = rgb_to_yuv_coef(pick_yuv_standard(h, yuv_standard));
    if ((h + 1) / 2 > 65535u)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: height %u too large", h);
    const bool any_size = nv12 || (w & 1) || (h & 1); // odd sizes (last column / row replicated) and NV12: one lane per chroma sample
    for (uint32_t done = 0; done < n; done += kMaxBatch) {
        const uint32_t m = std::min<uint32_t>(kMaxBatch, n - done);
        PlanesOut out{};
        PackedBatch in{};
        out.ys = i420_out->stride[0]; out.us = i420_out->stride[1]; out.vs = i420_out->stride[2];
        uint64_t a8 = out.ys, a4 = out.us | out.vs, ain = rgba_in->stride;
        for (uint32_t i = 0; i < m; i++) {
            out.y[i] = static_cast<uint8_t *>(outs[done + i].data[0]);
            out.u[i] = static_cast<uint8_t *>(outs[done + i].data[1]);
            out.v[i] = nv12 ? nullptr : static_cast<uint8_t *>(outs[done + i].data[2]);
            in.base[i] = static_cast<uint8_t *>(ins[done + i].data);
            a8 |= reinterpret_cast<uintptr_t>(out.y[i]);
            a4 |= reinterpret_cast<uintptr_t>(out.u[i]) | reinterpret_cast<uintptr_t>(out.v[i]);
            ain |= reinterpret_cast<uintptr_t>(in.base[i]);
        }
        const bool dword_ok = (ain & 3) == 0;
        if (any_size) {
            const dim3 grid_any(((w + 1) / 2 + kCvtBlock - 1) / kCvtBlock, (h + 1) / 2, m);
            if (nv12) MVFX_LAUNCH((rgba_to_yuv420_any_kernel<true>), grid_any, dim3(kCvtBlock), 0, st, in, (uint64_t)rgba_in->stride, w, h, k, out, dword_ok);
            else MVFX_LAUNCH((rgba_to_yuv420_any_kernel<false>), grid_any, dim3(kCvtBlock), 0, st, in, (uint64_t)rgba_in->stride, w, h, k, out, dword_ok);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        const bool aligned = (a8 & 7) == 0 && (a4 & 3) == 0 && (ain & 15) == 0;
        // eight columns per lane: with four (contiguous 16-byte loads, 2-byte chroma stores, the co-sited filter's left column read
        // again every 4 instead of every 8 pixels) the read-dominated direction does not move: 113.1 k vs 113.9 k fps
        constexpr int kCols = 8;
        const dim3 grid(((w + kCols - 1) / kCols + kCvtBlock - 1) / kCvtBlock, h / 2, m);
        if (aligned)
            MVFX_LAUNCH((rgba_to_i420_kernel<true, kCols>), grid, dim3(kCvtBlock), 0, st, in, (uint64_t)rgba_in->stride, w, h, k, out, dword_ok);
        else
            MVFX_LAUNCH((rgba_to_i420_kernel<false, kCols>), grid, dim3(kCvtBlock), 0, st, in, (uint64_t)rgba_in->stride, w, h, k, out, dword_ok);
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}

extern "C" {

int mvfx_convert_i420_to_rgba(const mvfx_planar_frame *i420_in, const mvfx_frame *rgba_out, int32_t yuv_standard, mvfx_stream stream)
{
    return i420_to_rgba_impl(i420_in, rgba_out, 1, yuv_standard, as_stream(stream));
}

int mvfx_convert_i420_to_rgba_frames(const mvfx_planar_frame *i420_in, const mvfx_frame *rgba_out, uint32_t n_frames,
                                     int32_t yuv_standard, mvfx_stream stream)
{
    return i420_to_rgba_impl(i420_in, rgba_out, n_frames, yuv_standard, as_stream(stream));
}

int mvfx_convert_rgba_to_i420(const mvfx_frame *rgba_in, const mvfx_planar_frame *i420_out, int32_t yuv_standard, mvfx_stream stream)
{
    return rgba_to_i420_impl(rgba_in, i420_out, 1, yuv_standard, as_stream(stream));
}

int mvfx_convert_rgba_to_i420_frames(const mvfx_frame *rgba_in, const mvfx_planar_frame *i420_out, uint32_t n_frames,
                                     int32_t yuv_standard, mvfx_stream stream)
{
    return rgba_to_i420_impl(rgba_in, i420_out, n_frames, yuv_standard, as_stream(stream));
}

int mvfx_convert_nv12_to_rgba(const mvfx_planar_frame *nv12_in, const mvfx_frame *rgba_out, int32_t yuv_standard, mvfx_stream stream)
{
    if (int rc = check_nv12(nv12_in, "convert input"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(rgba_out, "convert output"); rc != MVFX_OK) return rc;
    if (rgba_out->format != MVFX_FORMAT_RGBA)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "convert: output format %d is not RGBA", rgba_out->format);
    if (nv12_in->width != rgba_out->width || nv12_in->height != rgba_out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "convert: input %ux%u and output %ux%u differ", nv12_in->width, nv12_in->height, rgba_out->width, rgba_out->height);
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: yuv_standard %d is not 0 (by height), 1 (BT.601), 2 (BT.709), 3 (BT.2020)", yuv_standard);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t w = nv12_in->width, h = nv12_in->height;
    if (w == 0 || h == 0) return MVFX_OK;
    if (h > 65535u) return fail(MVFX_ERR_INVALID_ARGUMENT, "convert: height %u too large", h);
    const int std_ = pick_yuv_standard(h, yuv_standard);
    const YuvToRgbCoef k = yuv_to_rgb_coef(std_);
    const bool aligned = ((reinterpret_cast<uintptr_t>(rgba_out->data) | rgba_out->stride) & 15) == 0;
    const dim3 grid(((w + 3) / 4 + kCvtBlock - 1) / kCvtBlock, h);
    MVFX_LAUNCH(nv12_to_rgba_kernel, grid, dim3(kCvtBlock), 0, as_stream(stream), static_cast<const uint8_t *>(nv12_in->data[0]),
                       static_cast<const uint8_t *>(nv12_in->data[1]), (uint64_t)nv12_in->stride[0], (uint64_t)nv12_in->stride[1], w, h, k,
                       std_ != 1, static_cast<uint8_t *>(rgba_out->data), (uint64_t)rgba_out->stride, aligned);
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

int mvfx_convert_rgba_to_nv12(const mvfx_frame *rgba_in, const mvfx_planar_frame *nv12_out, int32_t yuv_standard, mvfx_stream stream)
{
    return rgba_to_i420_impl(rgba_in, nv12_out, 1, yuv_standard, as_stream(stream), true);
}

} // extern "C"
