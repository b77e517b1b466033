// Per-pixel arithmetic of GStreamer 1.14.0's `videoconvert` for I420 <-> RGBA with default caps -- the element the
// reference's colorlut example wraps around the filter (video/colorlut/src/colorlut/imp.rs:17-19).
// gst-plugins-base is not under the reference tree; oracle/convert_oracle.c restates the same arithmetic on the CPU and
// is pinned byte for byte against the real element (tests/golden/make_videoconvert_golden.py).
//
//  I420 -> RGBA  (video-converter.c convert_I420_pack_ARGB -> orc video_orc_convert_I420_BGRA):
//      w(c)  = splatbw(c - 128)                       the byte c^0x80 repeated in both halves of an int16
//      wy    = mulhsw(w(Y), p1)                        (a * b) >> 16
//      R     = convssswb(addssw(wy, mulhsw(w(V), p2))) + 128, B with w(U), p3, G with w(U), p4 and w(V), p5
//      The 16-bit saturating adds can never saturate (|wy| <= 148, the chroma terms <= 273), so
//      byte = clamp(wy + terms + 128, 0, 255): one v_med3_i32 per channel.
//  RGBA -> I420  (video_orc_matrix8 + video-chroma.c):  c = clamp(((a R + b G + c B) >> 8) + offset, 0, 255);
//      chroma averaged vertically first ((a + b + 1) >> 1), then horizontally: site none (a + b + 1) >> 1,
//      h-cosited (l + 2c + r + 2) >> 2 with (3a + b + 2) >> 2 at the first and (l + 3c + 2) >> 2 at the last sample.
//      All Y coefficients are positive bytes => one v_dot4_u32_u8; U / V = (positive dot - negative dot) >> 8.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mvfx {

struct YuvToRgbCoef {
    int32_t p1, p2, p3, p4, p5; // Y gain, V->R, U->B, U->G, V->G, all x256
};

struct RgbToYuvCoef {
    uint32_t y;             // bytes {R,G,B,0} coefficients of Y (all positive)
    uint32_t u_pos, u_neg;  // U = (dot(px, u_pos) - dot(px, u_neg)) >> 8 + 128
    uint32_t v_pos, v_neg;
    int32_t cosited;        // horizontal chroma filter: 0 = pair average, 1 = 1-2-1 co-sited
};

// standard: 1 BT.601 (site none), 2 BT.709 (h-cosited), 3 BT.2020 (h-cosited); 0 = GStreamer 1.14's default by height
__host__ __device__ inline int pick_yuv_standard(uint32_t height, int standard)
{
    if (standard >= 1 && standard <= 3) return standard;
    return height <= 576 ? 1 : (height < 2160 ? 2 : 3);
}

inline YuvToRgbCoef yuv_to_rgb_coef(int std_)
{
    switch (std_) {
    case 1: return {298, 409, 516, -100, -208};
    case 2: return {298, 459, 541, -55, -136};
    default: return {298, 430, 548, -48, -167};
    }
}

inline RgbToYuvCoef rgb_to_yuv_coef(int std_)
{
    auto pack = [](uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); };
    switch (std_) {
    case 1: return {pack(66, 129, 25), pack(0, 0, 112), pack(38, 74, 0), pack(112, 0, 0), pack(0, 94, 18), 0};
    case 2: return {pack(47, 157, 16), pack(0, 0, 112), pack(26, 87, 0), pack(112, 0, 0), pack(0, 102, 10), 1};
    default: return {pack(58, 149, 13), pack(0, 0, 112), pack(31, 81, 0), pack(112, 0, 0), pack(0, 103, 9), 1};
    }
}

// splatbw(c - 128) as a sign-extended int
__device__ __forceinline__ int32_t splat_s16(uint32_t byte_value)
{
    const uint32_t ub = byte_value ^ 0x80u;
    return (int32_t)(int16_t)((ub << 8) | ub);
}

// chroma terms shared by the (up to) four pixels of one chroma sample
struct ChromaTerms {
    int32_t r, g, b;
};

__device__ __forceinline__ ChromaTerms chroma_terms(uint32_t u, uint32_t v, const YuvToRgbCoef &k)
{
    // 16-bit x 11-bit signed products: v_mul_i32_i24 (full rate) is exact, a 32-bit v_mul_lo is quarter rate
    const int32_t wu = splat_s16(u), wv = splat_s16(v);
    return {__mul24(wv, k.p2) >> 16, (__mul24(wu, k.p4) >> 16) + (__mul24(wv, k.p5) >> 16), __mul24(wu, k.p3) >> 16};
}

__device__ __forceinline__ int32_t clamp_u8(int32_t v) { return min(max(v, 0), 255); }

// one RGBA pixel (alpha 255) from a luma byte and the chroma terms
__device__ __forceinline__ uint32_t yuv_pixel(uint32_t y, const ChromaTerms &c, const YuvToRgbCoef &k)
{
    const int32_t wy = (__mul24(splat_s16(y), k.p1) >> 16) + 128;
    return (uint32_t)clamp_u8(wy + c.r) | ((uint32_t)clamp_u8(wy + c.g) << 8) | ((uint32_t)clamp_u8(wy + c.b) << 16) | 0xff000000u;
}

__device__ __forceinline__ uint32_t rgb_luma(uint32_t px, const RgbToYuvCoef &k)
{
    return (uint32_t)clamp_u8((int32_t)(__builtin_amdgcn_udot4(px & 0x00ffffffu, k.y, 0u, false) >> 8) + 16);
}

__device__ __forceinline__ int32_t rgb_u(uint32_t px, const RgbToYuvCoef &k)
{
    const uint32_t q = px & 0x00ffffffu;
    const int32_t d = (int32_t)__builtin_amdgcn_udot4(q, k.u_pos, 0u, false) - (int32_t)__builtin_amdgcn_udot4(q, k.u_neg, 0u, false);
    return clamp_u8((d >> 8) + 128);
}

__device__ __forceinline__ int32_t rgb_v(uint32_t px, const RgbToYuvCoef &k)
{
    const uint32_t q = px & 0x00ffffffu;
    const int32_t d = (int32_t)__builtin_amdgcn_udot4(q, k.v_pos, 0u, false) - (int32_t)__builtin_amdgcn_udot4(q, k.v_neg, 0u, false);
    return clamp_u8((d >> 8) + 128);
}

// ---- a per-pixel RGBA filter applied to an I420 frame in one pass (videoconvert ! filter ! videoconvert fused) ----
// One lane owns an 8 x 2 pixel tile = 4 chroma samples, processed as four 2 x 2 blocks: decode (chroma terms shared by
// the block), px_fn on every RGBA pixel (alpha 255 in, alpha ignored out), luma bytes and the vertically averaged
// chroma of the 8 columns, then the horizontal chroma filter.  The co-sited filter (HD / UHD) needs the averaged
// chroma of the column LEFT of the tile: lanes hand their last column to their right neighbour through `edge` (LDS,
// kI420Block int2); the first lane of a workgroup evaluates that column itself (it reads INPUT pixels of the
// neighbouring workgroup's tile, so input and output planes must not alias).
// Requires width % 8 == 0, even height, luma rows 8-byte and chroma rows 4-byte aligned; blockDim.x == kI420Block,
// grid = (ceil(width / 8 / kI420Block), height / 2).
constexpr int kI420Block = 256;

struct I420Planes {
    const uint8_t *iy, *iu, *iv;
    uint8_t *oy, *ou, *ov;
    uint64_t iys, ius, ivs, oys, ous, ovs;
};

// COMPACT = false: a workgroup is a 2048 x 2 pixel strip (lane after lane along the row), grid = (ceil(width / 2048), height / 2).
// COMPACT = true:  a wave is a 64 x 16 pixel block (8 lanes across, 8 down), a workgroup four of them side by side (256 x 16),
//                  grid = (ceil(width / 256), ceil(height / 16)): the pixels of a wave are close to each other in the picture, which
//                  the wave-local LUT window of colorlut needs.
template <bool COMPACT>
__device__ __forceinline__ void i420_lane_origin(uint32_t &x0, uint32_t &y0, uint32_t &edge_index, bool &has_left_in_group)
{
    if constexpr (COMPACT) {
        const uint32_t wave = threadIdx.x >> 6, lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7;
        x0 = (blockIdx.x * 4 + wave) * 64 + lx * 8;
        y0 = blockIdx.y * 16 + ly * 2;
        edge_index = ly * 32 + wave * 8 + lx;
        has_left_in_group = (wave * 8 + lx) > 0;
    } else {
        x0 = (blockIdx.x * kI420Block + threadIdx.x) * 8;
        y0 = blockIdx.y * 2;
        edge_index = threadIdx.x;
        has_left_in_group = threadIdx.x > 0;
    }
}

template <bool COMPACT = false, typename F>
__device__ __forceinline__ void i420_fused_tile(const I420Planes &pl, uint32_t width, uint32_t height, const YuvToRgbCoef &kin,
                                                const RgbToYuvCoef &kout, int2 *edge, F &&px_fn)
{
    uint32_t x0, y0, edge_index;
    bool has_left;
    i420_lane_origin<COMPACT>(x0, y0, edge_index, has_left);
    const uint32_t crow = y0 / 2;
    const bool active = x0 < width && y0 < height;
    const bool cosited = kout.cosited != 0;
    int32_t cu[8], cv[8];
    uint32_t ya0 = 0, ya1 = 0, yb0 = 0, yb1 = 0;
    const uint8_t *yr0 = pl.iy + (uint64_t)y0 * pl.iys, *yr1 = yr0 + pl.iys;
    const uint8_t *ur = pl.iu + (uint64_t)crow * pl.ius, *vr = pl.iv + (uint64_t)crow * pl.ivs;
    if (active) {
        const uint2 ya = *reinterpret_cast<const uint2 *>(yr0 + x0), yb = *reinterpret_cast<const uint2 *>(yr1 + x0);
        const uint32_t u4 = *reinterpret_cast<const uint32_t *>(ur + x0 / 2), v4 = *reinterpret_cast<const uint32_t *>(vr + x0 / 2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const ChromaTerms c = chroma_terms((u4 >> (8 * j)) & 0xffu, (v4 >> (8 * j)) & 0xffu, kin);
            const uint32_t wa = j < 2 ? ya.x : ya.y, wb = j < 2 ? yb.x : yb.y;
            const int s = (2 * j & 3) * 8;
#pragma unroll
            for (int e = 0; e < 2; e++) { // the two columns of the block
                const uint32_t qa = px_fn(yuv_pixel((wa >> (s + 8 * e)) & 0xffu, c, kin));
                const uint32_t qb = px_fn(yuv_pixel((wb >> (s + 8 * e)) & 0xffu, c, kin));
                const uint32_t la = rgb_luma(qa, kout), lb = rgb_luma(qb, kout);
                const int col = 2 * j + e;
                if (col < 4) { ya0 |= la << (8 * col); yb0 |= lb << (8 * col); }
                else { ya1 |= la << (8 * (col - 4)); yb1 |= lb << (8 * (col - 4)); }
                cu[col] = (rgb_u(qa, kout) + rgb_u(qb, kout) + 1) >> 1; // vertical first
                cv[col] = (rgb_v(qa, kout) + rgb_v(qb, kout) + 1) >> 1;
            }
        }
        *reinterpret_cast<uint2 *>(pl.oy + (uint64_t)y0 * pl.oys + x0) = make_uint2(ya0, ya1);
        *reinterpret_cast<uint2 *>(pl.oy + (uint64_t)(y0 + 1) * pl.oys + x0) = make_uint2(yb0, yb1);
    }
    int32_t lu = 0, lv = 0;
    if (cosited) { // uniform branch
        edge[edge_index] = active ? make_int2(cu[7], cv[7]) : make_int2(0, 0);
        __syncthreads();
        if (active && x0 > 0) {
            if (has_left) {
                lu = edge[edge_index - 1].x;
                lv = edge[edge_index - 1].y;
            } else { // left neighbour lives in another workgroup: evaluate column x0 - 1 here
                const ChromaTerms c = chroma_terms(ur[(x0 - 1) / 2], vr[(x0 - 1) / 2], kin);
                const uint32_t qa = px_fn(yuv_pixel(yr0[x0 - 1], c, kin)), qb = px_fn(yuv_pixel(yr1[x0 - 1], c, kin));
                lu = (rgb_u(qa, kout) + rgb_u(qb, kout) + 1) >> 1;
                lv = (rgb_v(qa, kout) + rgb_v(qb, kout) + 1) >> 1;
            }
        }
    }
    if (!active) return;
    const uint32_t cw = width / 2;
    uint32_t u4o = 0, v4o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t ci = x0 / 2 + i;
        const int32_t l_u = i ? cu[2 * i - 1] : lu, l_v = i ? cv[2 * i - 1] : lv;
        uint32_t ru, rv;
        if (!cosited) { ru = (uint32_t)((cu[2 * i] + cu[2 * i + 1] + 1) >> 1); rv = (uint32_t)((cv[2 * i] + cv[2 * i + 1] + 1) >> 1); }
        else if (ci == 0) { ru = (uint32_t)((3 * cu[0] + cu[1] + 2) >> 2); rv = (uint32_t)((3 * cv[0] + cv[1] + 2) >> 2); }
        else if (ci == cw - 1) { ru = (uint32_t)((l_u + 3 * cu[2 * i] + 2) >> 2); rv = (uint32_t)((l_v + 3 * cv[2 * i] + 2) >> 2); }
        else { ru = (uint32_t)((l_u + 2 * cu[2 * i] + cu[2 * i + 1] + 2) >> 2); rv = (uint32_t)((l_v + 2 * cv[2 * i] + cv[2 * i + 1] + 2) >> 2); }
        u4o |= ru << (8 * i);
        v4o |= rv << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(pl.ou + (uint64_t)crow * pl.ous + x0 / 2) = u4o;
    *reinterpret_cast<uint32_t *>(pl.ov + (uint64_t)crow * pl.ovs + x0 / 2) = v4o;
}

} // namespace mvfx
