// colorlut: what the translation units of the element share -- the kernel parameter block, the device arithmetic of the reference's
// transform_rgba_* (video/colorlut/src/colorlut/imp.rs:399-543) in its literal and its strength-reduced ("FAST", finite domain) form, the
// geometry constants of the window kernels and their launchers.
//   colorlut_kernels.hip         the per-lane gather kernels (literal, FAST, LDS-staged cubes / tables, RGBA64, RGB10A2, the baked table),
//                                 the device copies of a LUT, the choice of kernel, the C ABI
//   colorlut_window_kernels.hip  the LDS window kernels for RGBA8 / RGBA64 through a 3-D LUT (cell window, x-prelerped per-wave window,
//                                 workgroup window), the content probe, the fused I420 forms
#pragma once

#include "mvfx_internal.h"
#include "convert_math.hpp"

namespace mvfx {

constexpr int kBlock = 256;
constexpr uint32_t kLds3dMaxSize = 21;   // 21^3 * 16 B = 148,176 B
constexpr uint32_t kLds1dMaxSize = 4096; // 3 * 4096 * 4 B = 48 KB
constexpr int kLdsBlock = 1024;
constexpr uint32_t kCellMaxSize = 65;    // 65^3 * 96 B = 26 MB; larger cubes keep the node layout only
constexpr uint32_t kCellF4 = 6;          // float4 per cell of the cell-packed table (96 bytes).  Padding cells to one 128-byte line was tried for
                                         // uniform-random colours and buys nothing: L1 fetches 64-byte blocks from L2 and a 96-byte cell always
                                         // covers exactly two of them (TCP_TCC_READ_REQ 2.36 -> 2.1 per pixel, the 4.6 MB table no longer fits
                                         // one XCD's 4 MiB L2: 11.3 k vs 12.0 k frames/s; profiles/r2/colorlut_random_floor.txt)

// Constants of the FAST kernels, passed as kernel arguments so they sit in SGPRs (32-bit VOP2
// encodings; see hsv_math.hpp for the instruction-class measurements).
struct LutFast {
    float c_lo, c_hi;   // 1/255 (or 1/65535) = c_hi + c_lo   (tools/prove_exact.c P8)
    float out_scale;    // 255 or 65535
    float pred_half;    // 0.49999997: round-half-away == trunc(v + pred_half)   (P10)
};

struct LutParams {
    LutFast fast;
    const float4 *cells;  // 3-D cell-packed copy (8 corners per cell) or nullptr
    const uint32_t *tile_tables; // colorlut_tile_kernel: coordinate tables + neighbourhood piece offsets
    const float4 *xtable; // colorlut_xtile_kernel: the x-prelerped table, addressed in 16-byte pieces (or nullptr)
    const uint2 *xcoord;  // colorlut_xtile_kernel: 512 x {cell index x row pitch, fraction bits} (g, then b)
    const uint2 *xcoord_wg; // colorlut_xwg_kernel: the same with its window's pitches (or nullptr: cube smaller than its window)
    const float4 *cube;   // 3-D nodes
    const float *t[3];    // 1-D tables
    uint32_t size;
    float size_m1;        // `size as f32 - 1.0` (imp.rs:408, :438)
    float scale[3], offset[3];
};

// f32::clamp(0.0, 1.0): NaN propagates (imp.rs:473, :478, :538, :542)
__device__ __forceinline__ float std_clamp01(float v)
{
    v = (v < 0.0f) ? 0.0f : v;
    v = (v > 1.0f) ? 1.0f : v;
    return v;
}

__device__ __forceinline__ float div255_exact(float x) // prove_exact P1
{
    const float c = 1.0f / 255.0f;
    const float q0 = x * c;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q0, x), c, q0);
}

__device__ __forceinline__ float div65535_exact(float x) // prove_exact P6
{
    const float c = 1.0f / 65535.0f;
    const float q0 = x * c;
    return __builtin_fmaf(__builtin_fmaf(-65535.0f, q0, x), c, q0);
}

// norm_comp / norm_comp_u16 (imp.rs:471-479) followed by `* (size as f32 - 1.0)`
template <bool WIDE>
__device__ __forceinline__ float lattice_coord(uint32_t value, float scale, float offset, float size_m1)
{
    const float v = WIDE ? div65535_exact((float)value) : div255_exact((float)value);
    return std_clamp01(v * scale + offset) * size_m1;
}

// `(x.floor() as usize).min(max_idx)`: NaN -> 0 (x is never negative here)
__device__ __forceinline__ uint32_t lattice_index(float x, uint32_t max_idx)
{
    const float f = floorf(x);
    const uint32_t i = (f == f) ? (uint32_t)__float2uint_rz(fmaxf(f, 0.0f)) : 0u;
    return min(i, max_idx);
}

// f32::round(): half away from zero.  v is in [0, 65535] or NaN.
__device__ __forceinline__ float round_half_away(float v)
{
    const float t = truncf(v);
    return (v - t >= 0.5f) ? t + 1.0f : t; // v - t is exact; NaN compares false and t is NaN
}

// float_to_u8 / float_to_u16 (imp.rs:537-543)
template <bool WIDE>
__device__ __forceinline__ uint32_t float_to_unorm(float v)
{
    const float r = round_half_away(std_clamp01(v) * (WIDE ? 65535.0f : 255.0f));
    return (r == r) ? (uint32_t)__float2uint_rz(r) : 0u; // NaN as u8 == 0
}

__device__ __forceinline__ float lerp(float a, float b, float t) { return a + (b - a) * t; } // imp.rs:528-535

__device__ __forceinline__ uint32_t bswap16(uint32_t v) { return ((v & 0xffu) << 8) | ((v >> 8) & 0xffu); }

// ---------------------------------------------------------------- FAST path (finite domain)
//
// Same values as the literal functions above through exact reductions: u8/255 and u16/65535 as
// mul+fmac (P8), the [0,1] clamps on the VOP3 clamp bit (domain scale/offset finite => no NaN
// before the LUT; NaN/inf LUT nodes still propagate through the lerps and the final clamp maps
// NaN to 0 exactly like `NaN as u8`), floor(x) as the truncating convert (x >= 0),
// round-half-away as trunc(v + 0.49999997) (P10).  3-D cubes up to 65^3 are read from a
// cell-packed copy (all 8 corners of a cell in 96 contiguous bytes: 1-2 cache lines per pixel
// instead of four, immediate offsets instead of 7 address computations, 25 % fewer L1 bytes).

__device__ __forceinline__ float lf_fmac_sv(float acc, float s, float v)
{
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(s), "v"(v));
    return acc;
}

__device__ __forceinline__ float lf_add_clamp(float a, float b) // clamp(a + b, 0, 1): NaN -> 0
{
    float r;
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// lattice coordinate of one channel: byte/word value as float -> (index, fraction)
__device__ __forceinline__ void lf_coord(float raw, const LutFast &k, float scale, float offset, float size_m1,
                                         uint32_t &i0, float &t)
{
    const float v = lf_fmac_sv(raw * k.c_lo, k.c_hi, raw);       // RN(raw / 255) or RN(raw / 65535)
    // (skipping `* scale + offset` + clamp for the default domain -- exact there: v * 1.0 == v, v + -0.0 == v -- through
    // a wave-uniform branch made uniform-random frames 15 % SLOWER: the branches split the scheduling regions)
    const float x = lf_add_clamp(v * scale, offset) * size_m1;   // norm_comp * (size - 1), in [0, size-1]
    i0 = (uint32_t)__float2uint_rz(x);                           // floor (x >= 0); <= size-1 by construction
    t = x - (float)i0;
}

__device__ __forceinline__ float lf_lerp(float a, float b, float t) { return a + (b - a) * t; }

// trilinear over the 8 corners c[0..7] = c000,c100,c010,c110,c001,c101,c011,c111; returns the
// clamped [0,1] channel values
// DIFF: the odd corners hold the x-differences c1-c0, c3-c2, ... (RN(b - a), formed once on the host when the cell
// table is packed: the same IEEE subtraction the lerp would do), so the four x-lerps are a + d * t
template <bool DIFF = false>
__device__ __forceinline__ void lf_trilinear(const float4 (&c)[8], float tx, float ty, float tz, float &r, float &g, float &b)
{
#define MVFX_LX(a, b_) (DIFF ? (a) + (b_) * tx : lf_lerp(a, b_, tx))
#define MVFX_CH(ch)                                                                              \
    {                                                                                            \
        const float c00 = MVFX_LX(c[0].ch, c[1].ch), c10 = MVFX_LX(c[2].ch, c[3].ch);            \
        const float c01 = MVFX_LX(c[4].ch, c[5].ch), c11 = MVFX_LX(c[6].ch, c[7].ch);            \
        const float c0 = lf_lerp(c00, c10, ty), c1 = lf_lerp(c01, c11, ty);                      \
        ch##_out = lf_add_clamp(c0, (c1 - c0) * tz);                                             \
    }
    float x_out, y_out, z_out;
    MVFX_CH(x) MVFX_CH(y) MVFX_CH(z)
#undef MVFX_CH
#undef MVFX_LX
    r = x_out; g = y_out; b = z_out;
}

// The 24 floats of the cell the lane used last: consecutive pixels of real pictures mostly fall into
// the same LUT cell (a 33^3 cell spans 8 byte values per axis), and a lane owns 4 (RGBA8) or 2 (RGBA64)
// consecutive pixels, so the 96-byte gather is skipped (exec-masked off) whenever the cell repeats.
struct CellCache {
    uint32_t index = 0xffffffffu;
    float f[24];
};

template <bool CELLS, typename CUBE>
__device__ __forceinline__ void lf_sample_3d(CUBE cube, const float4 *cells, uint32_t size, uint32_t x0, uint32_t y0,
                                             uint32_t z0, float tx, float ty, float tz, float &r, float &g, float &b,
                                             CellCache &cache)
{
    float4 c[8];
    if constexpr (CELLS) {
        // 96-byte cell: 8 corners x (r,g,b) f32, 3.45 MB for 33^3 (fits one XCD's 4 MiB L2)
        const uint32_t index = x0 + size * (y0 + size * z0);
        if (index != cache.index) {
            const float4 *cell = cells + (size_t)index * kCellF4;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const float4 v = cell[i];
                cache.f[4 * i] = v.x; cache.f[4 * i + 1] = v.y; cache.f[4 * i + 2] = v.z; cache.f[4 * i + 3] = v.w;
            }
            cache.index = index;
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            c[i] = make_float4(cache.f[3 * i], cache.f[3 * i + 1], cache.f[3 * i + 2], 0.0f);
    } else {
        const uint32_t m = size - 1;
        const uint32_t x1 = min(x0 + 1, m), y1 = min(y0 + 1, m), z1 = min(z0 + 1, m);
        const uint32_t s2 = size * size;
        const uint32_t r00 = y0 * size + z0 * s2, r10 = y1 * size + z0 * s2, r01 = y0 * size + z1 * s2, r11 = y1 * size + z1 * s2;
        c[0] = cube[x0 + r00]; c[1] = cube[x1 + r00]; c[2] = cube[x0 + r10]; c[3] = cube[x1 + r10];
        c[4] = cube[x0 + r01]; c[5] = cube[x1 + r01]; c[6] = cube[x0 + r11]; c[7] = cube[x1 + r11];
    }
    lf_trilinear<CELLS>(c, tx, ty, tz, r, g, b); // the cell-packed table stores x-differences in its odd corners
}

// RGBA8 pixel: converted channels are written into bytes 0..2 of the pixel register in place, so
// the alpha byte is carried over without a merge instruction.
template <bool IS3D, bool CELLS, typename CUBE, typename TABLE>
__device__ __forceinline__ uint32_t lf_px8(uint32_t px, const LutParams &p, CUBE cube, TABLE t0, TABLE t1, TABLE t2,
                                           CellCache &cache)
{
    // (a 3 x 256 LDS table of the per-byte (index, fraction) pairs was measured SLOWER than these 27 VALU
    // instructions: 31.5 k vs 36.2 k frames/s on the smpte frame -- random ds_read_b64 bank conflicts)
    uint32_t ix, iy, iz;
    float tx, ty, tz;
    lf_coord((float)(px & 0xffu), p.fast, p.scale[0], p.offset[0], p.size_m1, ix, tx);
    lf_coord((float)((px >> 8) & 0xffu), p.fast, p.scale[1], p.offset[1], p.size_m1, iy, ty);
    lf_coord((float)((px >> 16) & 0xffu), p.fast, p.scale[2], p.offset[2], p.size_m1, iz, tz);
    float r, g, b;
    if constexpr (IS3D) {
        lf_sample_3d<CELLS>(cube, p.cells, p.size, ix, iy, iz, tx, ty, tz, r, g, b, cache);
    } else {
        const uint32_t m = p.size - 1;
        const float a0 = t0[ix], b0 = t0[min(ix + 1, m)], a1 = t1[iy], b1 = t1[min(iy + 1, m)], a2 = t2[iz], b2 = t2[min(iz + 1, m)];
        r = lf_add_clamp(a0, (b0 - a0) * tx);
        g = lf_add_clamp(a1, (b1 - a1) * ty);
        b = lf_add_clamp(a2, (b2 - a2) * tz);
    }
    const float yr = r * p.fast.out_scale + p.fast.pred_half, yg = g * p.fast.out_scale + p.fast.pred_half,
                yb = b * p.fast.out_scale + p.fast.pred_half;
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yr));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yg));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yb));
    return px;
}

template <bool IS3D, bool CELLS, bool LE, typename CUBE, typename TABLE>
__device__ __forceinline__ void lf_px16(uint32_t &w0, uint32_t &w1, const LutParams &p, CUBE cube, TABLE t0, TABLE t1, TABLE t2,
                                        CellCache &cache)
{
    uint32_t rv = w0 & 0xffffu, gv = w0 >> 16, bv = w1 & 0xffffu;
    if constexpr (!LE) { rv = bswap16(rv); gv = bswap16(gv); bv = bswap16(bv); }
    uint32_t ix, iy, iz;
    float tx, ty, tz;
    lf_coord((float)rv, p.fast, p.scale[0], p.offset[0], p.size_m1, ix, tx);
    lf_coord((float)gv, p.fast, p.scale[1], p.offset[1], p.size_m1, iy, ty);
    lf_coord((float)bv, p.fast, p.scale[2], p.offset[2], p.size_m1, iz, tz);
    float r, g, b;
    if constexpr (IS3D) {
        lf_sample_3d<CELLS>(cube, p.cells, p.size, ix, iy, iz, tx, ty, tz, r, g, b, cache);
    } else {
        const uint32_t m = p.size - 1;
        const float a0 = t0[ix], b0 = t0[min(ix + 1, m)], a1 = t1[iy], b1 = t1[min(iy + 1, m)], a2 = t2[iz], b2 = t2[min(iz + 1, m)];
        r = lf_add_clamp(a0, (b0 - a0) * tx);
        g = lf_add_clamp(a1, (b1 - a1) * ty);
        b = lf_add_clamp(a2, (b2 - a2) * tz);
    }
    uint32_t ro = (uint32_t)__float2uint_rz(r * p.fast.out_scale + p.fast.pred_half);
    uint32_t go = (uint32_t)__float2uint_rz(g * p.fast.out_scale + p.fast.pred_half);
    uint32_t bo = (uint32_t)__float2uint_rz(b * p.fast.out_scale + p.fast.pred_half);
    if constexpr (!LE) { ro = bswap16(ro); go = bswap16(go); bo = bswap16(bo); }
    w0 = ro | (go << 16);
    w1 = bo | (w1 & 0xffff0000u);
}

// FAST row walker: aligned 16-byte vectors only (the launcher falls back to the literal kernels otherwise)
template <bool IS3D, bool CELLS, bool WIDE, bool LE, typename CUBE, typename TABLE>
__device__ __forceinline__ void lf_rows(const uint8_t *in, uint8_t *out, uint64_t width, uint32_t rows, uint64_t in_stride,
                                        uint64_t out_stride, const LutParams &p, CUBE cube, TABLE t0, TABLE t1, TABLE t2,
                                        uint32_t first_group, uint32_t group_stride, uint32_t first_row, uint32_t row_stride)
{
    constexpr uint32_t PXV = WIDE ? 2 : 4;
    constexpr uint32_t BPP = WIDE ? 8 : 4;
    CellCache cache;
    for (uint32_t row = first_row; row < rows; row += row_stride) {
        const uint8_t *iline = in + (uint64_t)row * in_stride;
        uint8_t *oline = out + (uint64_t)row * out_stride;
        const uint64_t groups = (width + PXV - 1) / PXV;
        for (uint64_t g = first_group; g < groups; g += group_stride) {
            const uint64_t x = g * PXV;
            if (x + PXV <= width) {
                uint4 v = *reinterpret_cast<const uint4 *>(iline + x * BPP);
                if constexpr (WIDE) {
                    lf_px16<IS3D, CELLS, LE>(v.x, v.y, p, cube, t0, t1, t2, cache);
                    lf_px16<IS3D, CELLS, LE>(v.z, v.w, p, cube, t0, t1, t2, cache);
                } else {
                    v.x = lf_px8<IS3D, CELLS>(v.x, p, cube, t0, t1, t2, cache);
                    v.y = lf_px8<IS3D, CELLS>(v.y, p, cube, t0, t1, t2, cache);
                    v.z = lf_px8<IS3D, CELLS>(v.z, p, cube, t0, t1, t2, cache);
                    v.w = lf_px8<IS3D, CELLS>(v.w, p, cube, t0, t1, t2, cache);
                }
                *reinterpret_cast<uint4 *>(oline + x * BPP) = v;
            } else {
                for (uint64_t xx = x; xx < width; xx++) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(iline + xx * BPP);
                    uint32_t *o = reinterpret_cast<uint32_t *>(oline + xx * BPP);
                    if constexpr (WIDE) {
                        uint32_t w0 = q[0], w1 = q[1];
                        lf_px16<IS3D, CELLS, LE>(w0, w1, p, cube, t0, t1, t2, cache);
                        o[0] = w0; o[1] = w1;
                    } else {
                        o[0] = lf_px8<IS3D, CELLS>(q[0], p, cube, t0, t1, t2, cache);
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------- geometry of the window kernels (colorlut_window_kernels.hip)
// Shared with the host side: ensure_uploaded builds the per-byte coordinate tables with the LDS pitches premultiplied.
constexpr uint32_t kCoordEntries = 3 * 256;     // colorlut_tile_kernel: {cell index, fraction} per channel and byte value
constexpr uint32_t kXRowPieces = 384;           // 16-byte pieces per (y, z) row of the x-prelerped table: 256 entries x 24 B
// colorlut_xtile_kernel: a wave's window is kXRW r bytes x kXNY y cells x (kXNZ + 1) z rows of 24-byte entries.  24 r bytes in rounds 3 and 4.
// Round 5: what the kernel is short of is waves -- with two-pixel row passes (56-60 VGPRs instead of 94) 18 r bytes x 3 x 4 rows = 5 184 bytes
// per wave put six workgroups on a CU.  16 x 4K per launch, noise +-0 / 3 / 5 / 8:
//   24 r bytes, four-pixel passes (round 4) 78.4 / 75.4 / 70.0 / 54.8 k fps      24, two-pixel passes (five workgroups) 75.4 / 73.0 / 69.8 / 55.0
//   18, two-pixel passes (six) 80.5 / 78.1 / 69.9 / 41.1        16 (seven) 83.0 / 78.3 / 62.5 / 39.1        12 (eight) 83.7 / 60.7 / 41.0 / 36.9
// (busy pictures go to colorlut_xwg_kernel: the content probe; profiles/r5/colorlut_experiments.txt, section 9)
constexpr uint32_t kXRW = 18;
constexpr uint32_t kXNY = 3, kXNZ = 3, kXNZR = kXNZ + 1;
constexpr uint32_t kXPitchZ = kXRW * 24, kXPitchY = kXNZR * kXPitchZ; // LDS bytes between z rows / y cells of a window
// Rows of four pixels per lane: a wave's block is 64 x (4 x rows) pixels.  Round 4, 16 x 4K natural-like frames: 8 / 12 / 16 / 20 / 24 rows of pixels
// 60.5 / 68.9 / 74.4 / 75.5 / 73.6 k fps.
constexpr uint32_t kXRows = 5;
// colorlut_xwg_kernel: ONE window per workgroup (2 x 2 waves, a 128 x 40 block): r bytes, y cells, z cells, z rows
constexpr uint32_t kWgRW = 38, kWgNY = 5, kWgNZ = 5, kWgNZR = kWgNZ + 1;
constexpr uint32_t kWgPitchZ = kWgRW * 24, kWgPitchY = kWgNZR * kWgPitchZ, kWgWinBytes = kWgNY * kWgPitchY;
// the content probe runs in front of every kProbeEvery-th call on a LUT
constexpr uint32_t kProbeEvery = 32;

// launchers (colorlut_window_kernels.hip); every kernel goes out through MVFX_LAUNCH except the probe, which is no part of a frame's work
void launch_colorlut_xtable_build(const float4 *cube, const uint32_t *tile_tables, uint32_t size, float *xtable, uint64_t entries);
void launch_colorlut_xtile(dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width, uint32_t height, uint32_t in_stride,
                           uint32_t out_stride, const LutParams &p);
void launch_colorlut_xwg(dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width, uint32_t height, uint32_t in_stride,
                         uint32_t out_stride, const LutParams &p);
void launch_colorlut_probe(hipStream_t st, const uint8_t *frame, uint32_t width, uint32_t height, uint32_t stride, uint32_t *verdict);
// the cell-window kernel: wide = RGBA64 (le: little endian); narrow = 32 x 16 blocks instead of 64 x 16 (RGBA8 only)
void launch_colorlut_tile(bool wide, bool le, bool narrow, dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width,
                          uint32_t height, uint32_t in_stride, uint32_t out_stride, const LutParams &p);
void launch_colorlut_i420_window(bool xtile, dim3 grid, hipStream_t st, const I420Planes &pl, uint32_t width, uint32_t height, const LutParams &p,
                                 const YuvToRgbCoef &kin, const RgbToYuvCoef &kout);

} // namespace mvfx
