// colorlut kernels for gfx950 + C ABI (parsing lives in host/cube_parser.cpp).
//
// Replaces video/colorlut/src/colorlut/imp.rs:226-543:
//   transform_rgba_{1d,3d}, transform_rgba64_{1d,3d}<LE>, apply_*, norm_comp*, sample_1d,
//   sample_3d, lerp4, float_to_u8/u16.
// Arithmetic is transcribed literally (one rounding per op, no FMA contraction, std clamp with
// NaN propagation, round-half-away, saturating casts); only u8/255 and u16/65535 use the
// mul+fma+fma form proven exact in tools/prove_exact.c (P1, P6).
//
// Data layout: the 3-D cube stays in the reference's layout ([r,g,b,1.0] float4 per node,
// R fastest, parser.rs:43-53) so a corner is ONE aligned 16-byte gather and the x0/x1 pair of a
// cell is 32 contiguous bytes.  A 33^3 cube is 575 KB: it does not fit the 160 KB LDS
//  but is resident in every XCD's 4 MiB L2 after first touch.  Cubes with
// size <= 21 (148 KB as float4) are staged in LDS instead (one 1024-thread workgroup per CU
// walking the frame), which turns the 8 gathers into ds_read_b128.
// 1-D tables (<= 65536 entries x 3) are read from global/L2; tables with size <= 4096 are staged
// in LDS.
// Pixels: RGBA8 -> one lane owns 4 pixels (16 B in, 16 B out); RGBA64 -> 2 pixels (16 B).
#include "mvfx_internal.h"

#include "cube_parser.h"
#include "convert_math.hpp"

#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

struct mvfx_cube_lut {
    mvfx::CubeLut lut;
    std::mutex mu;
    int device = -1;       // device the copies below live on
    float *d_rgba = nullptr;
    uint32_t *d_tile_tables = nullptr; // tile kernel: 3 x 256 x (cell index, fraction) per byte value + 192 neighbourhood piece offsets
    uint32_t *d_xcoord = nullptr; // colorlut_xtile_kernel: per byte value of g and b {cell index x LDS row pitch, fraction bits}
    uint32_t *d_xcoord_wg = nullptr; // colorlut_xwg_kernel: the same with its window's pitches (cubes of 5+ points)
    // Content probe (round 5, see colorlut_probe_kernel): which of the two window kernels the automatic choice takes for this LUT's frames.
    // The probe kernel writes its verdict into a page-locked host word; the launcher reads it without synchronising (a verdict a few
    // launches old is as good: pictures of a stream resemble their predecessors) -- advisory state, both kernels produce the same bytes.
    uint32_t *h_probe = nullptr;            // [0]: 0 = no verdict yet, 1 = calm, 2 = busy; [1]: busy blocks of the last probe (of 256)
    std::atomic<uint32_t> probe_calls{0};
    float *d_xtable = nullptr; // x-prelerped table of colorlut_xtile_kernel: [y][z][r byte] x (X.rgb, D.rgb) f32 = 24 B (3-D, 4 <= size <= kCellMaxSize)
    float *d_cells = nullptr; // cell-packed copy: size^3 cells x 8 corners x (r,g,b) f32 = 96 B (3-D, size <= kCellMaxSize)
    float *d_table[3] = {nullptr, nullptr, nullptr};
    // baked table (placement 6): the LUT applied to every one of the 2^24 RGB byte triples, 64 MiB, entry = output R | G << 8 | B << 16
    // at index r | g << 8 | b << 16 -- produced by running this file's own interpolating kernels once over a 4096 x 4096 frame that
    // holds every colour, so its bytes are theirs by construction
    std::mutex bake_mu;
    uint32_t *d_baked = nullptr;
};

namespace mvfx {
namespace {

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

// sample_1d (imp.rs:482-490); TABLE is a global or LDS pointer
template <typename TABLE>
__device__ __forceinline__ float sample_1d(TABLE lut, uint32_t max_idx, float x)
{
    const uint32_t x0 = lattice_index(x, max_idx);
    const uint32_t x1 = min(x0 + 1, max_idx);
    const float t = x - (float)x0;
    const float a = lut[x0], b = lut[x1];
    return a + (b - a) * t;
}

// sample_3d (imp.rs:493-526), RGB lanes only (the alpha lane of lerp4 is never read)
template <typename CUBE>
__device__ __forceinline__ void sample_3d(CUBE cube, uint32_t size, float x, float y, float z,
                                          float &r, float &g, float &b)
{
    const uint32_t max_idx = size - 1;
    const uint32_t x0 = lattice_index(x, max_idx), y0 = lattice_index(y, max_idx), z0 = lattice_index(z, max_idx);
    const uint32_t x1 = min(x0 + 1, max_idx), y1 = min(y0 + 1, max_idx), z1 = min(z0 + 1, max_idx);
    const float tx = x - (float)x0, ty = y - (float)y0, tz = z - (float)z0;
    const uint32_t s2 = size * size;
    const uint32_t r00 = y0 * size + z0 * s2, r10 = y1 * size + z0 * s2;
    const uint32_t r01 = y0 * size + z1 * s2, r11 = y1 * size + z1 * s2;
    const float4 c000 = cube[x0 + r00], c100 = cube[x1 + r00];
    const float4 c010 = cube[x0 + r10], c110 = cube[x1 + r10];
    const float4 c001 = cube[x0 + r01], c101 = cube[x1 + r01];
    const float4 c011 = cube[x0 + r11], c111 = cube[x1 + r11];
#define MVFX_TRI(ch)                                                                           \
    lerp(lerp(lerp(c000.ch, c100.ch, tx), lerp(c010.ch, c110.ch, tx), ty),                     \
         lerp(lerp(c001.ch, c101.ch, tx), lerp(c011.ch, c111.ch, tx), ty), tz)
    r = MVFX_TRI(x);
    g = MVFX_TRI(y);
    b = MVFX_TRI(z);
#undef MVFX_TRI
}

__device__ __forceinline__ uint32_t bswap16(uint32_t v) { return ((v & 0xffu) << 8) | ((v >> 8) & 0xffu); }

// One RGBA8 pixel (dword) through the LUT; alpha byte copied (imp.rs:262, :291)
template <bool IS3D, typename CUBE, typename TABLE>
__device__ __forceinline__ uint32_t lut_px8(uint32_t px, const LutParams &p, CUBE cube, TABLE t0, TABLE t1, TABLE t2)
{
    const float x = lattice_coord<false>(px & 0xffu, p.scale[0], p.offset[0], p.size_m1);
    const float y = lattice_coord<false>((px >> 8) & 0xffu, p.scale[1], p.offset[1], p.size_m1);
    const float z = lattice_coord<false>((px >> 16) & 0xffu, p.scale[2], p.offset[2], p.size_m1);
    float r, g, b;
    if constexpr (IS3D) {
        sample_3d(cube, p.size, x, y, z, r, g, b);
    } else {
        r = sample_1d(t0, p.size - 1, x);
        g = sample_1d(t1, p.size - 1, y);
        b = sample_1d(t2, p.size - 1, z);
    }
    return float_to_unorm<false>(r) | (float_to_unorm<false>(g) << 8) | (float_to_unorm<false>(b) << 16) |
           (px & 0xff000000u);
}

// One RGBA64 pixel (two dwords: [r,g] [b,a]); per-sample endian swap, alpha word copied raw
template <bool IS3D, bool LE, typename CUBE, typename TABLE>
__device__ __forceinline__ void lut_px16(uint32_t &w0, uint32_t &w1, const LutParams &p, CUBE cube, TABLE t0, TABLE t1, TABLE t2)
{
    uint32_t rv = w0 & 0xffffu, gv = w0 >> 16, bv = w1 & 0xffffu;
    if constexpr (!LE) { rv = bswap16(rv); gv = bswap16(gv); bv = bswap16(bv); }
    const float x = lattice_coord<true>(rv, p.scale[0], p.offset[0], p.size_m1);
    const float y = lattice_coord<true>(gv, p.scale[1], p.offset[1], p.size_m1);
    const float z = lattice_coord<true>(bv, p.scale[2], p.offset[2], p.size_m1);
    float r, g, b;
    if constexpr (IS3D) {
        sample_3d(cube, p.size, x, y, z, r, g, b);
    } else {
        r = sample_1d(t0, p.size - 1, x);
        g = sample_1d(t1, p.size - 1, y);
        b = sample_1d(t2, p.size - 1, z);
    }
    uint32_t ro = float_to_unorm<true>(r), go = float_to_unorm<true>(g), bo = float_to_unorm<true>(b);
    if constexpr (!LE) { ro = bswap16(ro); go = bswap16(go); bo = bswap16(bo); }
    w0 = ro | (go << 16);
    w1 = bo | (w1 & 0xffff0000u);
}

// Processes row `row` of a frame; one lane = 16 bytes (4 RGBA8 or 2 RGBA64 pixels) when VEC.
template <bool IS3D, bool WIDE, bool LE, bool VEC, typename CUBE, typename TABLE>
__device__ __forceinline__ void lut_rows(const uint8_t *in, uint8_t *out, uint64_t width, uint32_t rows,
                                         uint64_t in_stride, uint64_t out_stride, const LutParams &p,
                                         CUBE cube, TABLE t0, TABLE t1, TABLE t2, uint32_t first_group,
                                         uint32_t group_stride, uint32_t first_row, uint32_t row_stride)
{
    constexpr uint32_t PXV = WIDE ? 2 : 4; // pixels per 16-byte vector
    constexpr uint32_t BPP = WIDE ? 8 : 4;
    for (uint32_t row = first_row; row < rows; row += row_stride) {
        const uint8_t *iline = in + (uint64_t)row * in_stride;
        uint8_t *oline = out + (uint64_t)row * out_stride;
        if constexpr (VEC) {
            const uint64_t groups = (width + PXV - 1) / PXV;
            for (uint64_t g = first_group; g < groups; g += group_stride) {
                const uint64_t x = g * PXV;
                if (x + PXV <= width) {
                    uint4 v = *reinterpret_cast<const uint4 *>(iline + x * BPP);
                    if constexpr (WIDE) {
                        lut_px16<IS3D, LE>(v.x, v.y, p, cube, t0, t1, t2);
                        lut_px16<IS3D, LE>(v.z, v.w, p, cube, t0, t1, t2);
                    } else {
                        v.x = lut_px8<IS3D>(v.x, p, cube, t0, t1, t2);
                        v.y = lut_px8<IS3D>(v.y, p, cube, t0, t1, t2);
                        v.z = lut_px8<IS3D>(v.z, p, cube, t0, t1, t2);
                        v.w = lut_px8<IS3D>(v.w, p, cube, t0, t1, t2);
                    }
                    *reinterpret_cast<uint4 *>(oline + x * BPP) = v;
                } else {
                    for (uint64_t xx = x; xx < width; xx++) {
                        const uint32_t *q = reinterpret_cast<const uint32_t *>(iline + xx * BPP);
                        uint32_t *o = reinterpret_cast<uint32_t *>(oline + xx * BPP);
                        if constexpr (WIDE) {
                            uint32_t w0 = q[0], w1 = q[1];
                            lut_px16<IS3D, LE>(w0, w1, p, cube, t0, t1, t2);
                            o[0] = w0; o[1] = w1;
                        } else {
                            o[0] = lut_px8<IS3D>(q[0], p, cube, t0, t1, t2);
                        }
                    }
                }
            }
        } else { // byte-granular fallback for unaligned planes
            for (uint64_t x = first_group; x < width; x += group_stride) {
                const uint8_t *q = iline + x * BPP;
                uint8_t *o = oline + x * BPP;
                if constexpr (WIDE) {
                    uint32_t w0 = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
                    uint32_t w1 = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
                    lut_px16<IS3D, LE>(w0, w1, p, cube, t0, t1, t2);
                    for (int i = 0; i < 4; i++) { o[i] = (uint8_t)(w0 >> (8 * i)); o[4 + i] = (uint8_t)(w1 >> (8 * i)); }
                } else {
                    const uint32_t px = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
                    const uint32_t r = lut_px8<IS3D>(px, p, cube, t0, t1, t2);
                    for (int i = 0; i < 4; i++) o[i] = (uint8_t)(r >> (8 * i));
                }
            }
        }
    }
}

// LUT read from global memory (L2-resident)
template <bool IS3D, bool WIDE, bool LE, bool VEC>
__global__ __launch_bounds__(kBlock) void colorlut_global_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width,
                                                                 uint32_t rows, uint64_t in_stride,
                                                                 uint64_t out_stride, LutParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z]; // one frame pair of the batch per grid z
    uint8_t *out = out_fb.base[blockIdx.z];
    lut_rows<IS3D, WIDE, LE, VEC>(in, out, width, rows, in_stride, out_stride, p, p.cube, p.t[0], p.t[1], p.t[2],
                                  blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, blockIdx.y, gridDim.y);
}

// LUT staged in LDS: persistent 1024-thread workgroups (one per CU) walk the frame
template <bool IS3D, bool WIDE, bool LE, bool VEC>
__global__ __launch_bounds__(kLdsBlock) void colorlut_lds_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width,
                                                                 uint32_t rows, uint64_t in_stride,
                                                                 uint64_t out_stride, LutParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z]; // one frame pair of the batch per grid z
    uint8_t *out = out_fb.base[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    if constexpr (IS3D) {
        float4 *cube = reinterpret_cast<float4 *>(lds_raw);
        const uint32_t n = p.size * p.size * p.size;
        for (uint32_t i = threadIdx.x; i < n; i += kLdsBlock)
            cube[i] = p.cube[i];
        __syncthreads();
        lut_rows<IS3D, WIDE, LE, VEC>(in, out, width, rows, in_stride, out_stride, p, (const float4 *)cube,
                                      (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                                      blockIdx.x * kLdsBlock + threadIdx.x, gridDim.x * kLdsBlock, blockIdx.y, gridDim.y);
    } else {
        float *t = reinterpret_cast<float *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < p.size; i += kLdsBlock) {
            t[i] = p.t[0][i];
            t[p.size + i] = p.t[1][i];
            t[2 * p.size + i] = p.t[2][i];
        }
        __syncthreads();
        lut_rows<IS3D, WIDE, LE, VEC>(in, out, width, rows, in_stride, out_stride, p, (const float4 *)nullptr,
                                      (const float *)t, (const float *)(t + p.size), (const float *)(t + 2 * p.size),
                                      blockIdx.x * kLdsBlock + threadIdx.x, gridDim.x * kLdsBlock, blockIdx.y, gridDim.y);
    }
}


// ---------------------------------------------------------------- RGB10A2_LE (d3d12colorlut's third format)
// d3d12colorlut accepts RGBA64_LE, RGB10A2_LE and RGBA on D3D12 memory (d3d12colorlut/imp.rs:236-244) and samples the LUT in
// an HLSL shader -- hardware filtering, not bit-defined.  Here the CPU element's arithmetic is extended the way its 8- and
// 16-bit paths are written (imp.rs:471-479, 537-543): v / 1023.0, the same clamp / lattice / trilinear steps,
// (clamp(v, 0, 1) * 1023.0).round(), the two alpha bits copied.  Little-endian dword: R bits 0-9, G 10-19, B 20-29, A 30-31.
template <bool IS3D>
__global__ __launch_bounds__(kBlock) void colorlut_rgb10a2_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t rows,
                                                                  uint64_t in_stride, uint64_t out_stride, LutParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z];
    uint8_t *out = out_fb.base[blockIdx.z];
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y)
        for (uint32_t x = blockIdx.x * kBlock + threadIdx.x; x < width; x += gridDim.x * kBlock) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(in + (uint64_t)row * in_stride + (uint64_t)x * 4);
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; c++)
                v[c] = std_clamp01((float)((w >> (10 * c)) & 1023u) / 1023.0f * p.scale[c] + p.offset[c]) * p.size_m1;
            float o[3];
            if constexpr (IS3D) {
                sample_3d(p.cube, p.size, v[0], v[1], v[2], o[0], o[1], o[2]);
            } else {
                o[0] = sample_1d(p.t[0], p.size - 1, v[0]);
                o[1] = sample_1d(p.t[1], p.size - 1, v[1]);
                o[2] = sample_1d(p.t[2], p.size - 1, v[2]);
            }
            uint32_t res = w & 0xC0000000u;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float r = round_half_away(std_clamp01(o[c]) * 1023.0f);
                res |= ((r == r) ? (uint32_t)__float2uint_rz(r) : 0u) << (10 * c);
            }
            *reinterpret_cast<uint32_t *>(out + (uint64_t)row * out_stride + (uint64_t)x * 4) = res;
        }
}

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


// ---------------------------------------------------------------- RGBA8 through a 3-D LUT, wave-local cell neighbourhood in LDS
//
// What bounds the per-lane gather kernel above is the vector L1's tag look-up rate, on natural content as much as on random
// content (rocprofv3, profiles/r2/colorlut_counters_*_before.txt: TCP busy 97 %, 3.2 / 6.1 look-ups per pixel at ~1.2 per
// clock per CU; VALUBusy 64 % / 28 %): every lane that needs a cell pays 6 look-ups for its 96 bytes, whoever else in the
// wave wants the same bytes.  Pictures are locally coherent in colour: this kernel gives a wave a compact pixel BLOCK (64 x 16 or 32 x 16)
// (spatially compact, unlike 256 consecutive pixels of a row), takes the LUT cell of the tile's centre pixel as anchor and
// loads the 3 x 3 x 3 cells around it -- nine runs of 288 contiguous bytes, 162 coalesced 16-byte pieces in three wave
// loads, ~54 look-ups -- into the wave's 2.6 KB of LDS.  A pixel whose cell lies in that neighbourhood (a cell of a 33^3
// cube spans 8 code values per axis, the window 24) reads its 24 floats with six ds_read_b128 (lanes on one cell
// broadcast); the others gather from the cell table in global memory/L2 exactly as before.  Same arithmetic (lf_coord,
// lf_trilinear<true>): bit-identical results.
constexpr int kTileNbCells = 27;                       // 3 x 3 x 3 cells
constexpr int kTileNbPieces = kTileNbCells * 6;        // 16-byte pieces
// LDS pitch of a window cell in 16-byte pieces.  Round 3: 7 (112 bytes), not 6: a ds_read_b128 serves 16 lanes at a time, a
// 16-byte piece covers 4 of the 64 banks, so piece i of cell n sits on bank group (pitch * n + i) mod 16 -- with pitch 6 the cells n
// and n + 8 of the 27 (e.g. the x-neighbour and the z-neighbour of the centre cell, dx - 1 against dz - 1) share their banks and
// lanes of one group that want both serialise (profiles/r2/colorlut_block_counters.txt: SQ_LDS_BANK_CONFLICT 2.6e7 of 9.3e7 LDS
// cycles per 16-frame launch, the LDS busy 56 % of the launch); with pitch 7 only cells 16 apart collide, which are never neighbours.
#ifndef MVFX_TILE_CELL_PITCH
#define MVFX_TILE_CELL_PITCH 7
#endif
constexpr int kTileCellPitch = MVFX_TILE_CELL_PITCH;
#ifndef MVFX_XTILE_RW
#define MVFX_XTILE_RW 18 // r bytes per window row of colorlut_xtile_kernel.  24 in rounds 3 and 4; round 5 (with two-pixel row passes, below): what the
                         // kernel is short of is waves -- 94 VGPRs and 32 KB of LDS per workgroup allowed five per SIMD.  Two pixels per pass
                         // need 56-60 VGPRs, and 18 r bytes x 3 x 4 rows = 5 184 bytes per wave put six workgroups on a CU:
                         //   16 x 4K per launch, noise +-0 / 3 / 5 / 8:  24 r bytes, four-pixel passes (round 4)  78.4 / 75.4 / 70.0 / 54.8 k fps
                         //                                             24, two-pixel passes (five workgroups)    75.4 / 73.0 / 69.8 / 55.0
                         //                                             18, two-pixel passes (six)                80.5 / 78.1 / 69.9 / 41.1
                         //                                             16 (seven)  83.0 / 78.3 / 62.5 / 39.1     12 (eight)  83.7 / 60.7 / 41.0 / 36.9
                         // (busy pictures go to colorlut_xwg_kernel: the content probe; profiles/r5/colorlut_experiments.txt, section 9)
#endif
constexpr int kTileWaveLdsFloat4 = kTileNbCells * kTileCellPitch + 2;  // +32 bytes: de-phases the four waves' regions over the banks

// The lattice coordinate of a channel depends on its byte value only: (cell index, fraction) come from a 3 x 256 entry
// table in LDS (built on the host with the same f32 steps, ensure_uploaded) instead of 7 VALU instructions per channel --
// the kernel is VALU-bound once the gathers are gone (rocprofv3: VALUBusy 100 %, profiles/r2/colorlut_tile_counters.txt).
// (Typed buffer loads for u8/255 and several tiles per wave were tried and measured slower here: -4 % and -7 %.  A 5 x 5 x 5
// window for big cubes -- a cell of a 65^3 cube spans only 4 code values, natural-like 4K frame 37.7 us against 24.0 us with
// 33^3 -- costs more than its hits save: 12 KB of cells per tile in twelve wave loads, 48 KB of LDS per workgroup; 65^3
// natural 54.4 us, flat bars 57 us against 30 us, and 33^3 natural 52 us.)
constexpr uint32_t kCoordEntries = 3 * 256;

// The 24 floats of the cell (ix, iy, iz): from the wave's LDS window when the cell lies in it, otherwise this lane's own gather
// from the cell table in global memory / L2 (six 16-byte loads).  A quad-cooperative form of the gather (the four lanes of a
// quad fetch one cell with two coalesced loads and hand it over through LDS: 2.8 instead of 6.1 L1 look-ups per pixel) was
// built and measured: no faster on uniform-random colours -- there the L1's miss path is the floor (a cell is two 64-byte L2
// requests, ~0.39 requests per clock per CU) -- and slower when only a few pixels of a tile fall outside
// (profiles/r2/colorlut_random_floor.txt).
// `nbr_base` is an LDS-address-space pointer on purpose: through a generic pointer the six reads become flat loads (the
// kernel then runs at 60 % of its speed).
typedef const __attribute__((address_space(3))) char *lds_bytes_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4_t *lds_float4_t;

__device__ __forceinline__ void tile_cell(lds_bytes_t nbr_base, uint32_t wave_lds_bytes, const LutParams &p, uint32_t ix, uint32_t iy,
                                          uint32_t iz, uint32_t ax, uint32_t ay, uint32_t az, float4 (&c)[8])
{
    const uint32_t dx = ix - ax, dy = iy - ay, dz = iz - az; // unsigned: below the anchor wraps to a huge value
    float4 c6[6];
    if (dx < 3u && dy < 3u && dz < 3u) {
        // 24-bit multiply-adds, the last one spelled out: plain `mine + index * 6` compiles to three quarter-rate v_mad_u64_u32
        const uint32_t nbi = __umul24(dz, 9u) + __umul24(dy, 3u) + dx;
        uint32_t off;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(nbi), "s"((uint32_t)(kTileCellPitch * 16)), "v"(wave_lds_bytes));
        lds_float4_t cell = (lds_float4_t)(nbr_base + off);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const f32x4_t v = cell[i];
            c6[i] = make_float4(v.x, v.y, v.z, v.w);
        }
    } else {
        const float4 *cell = p.cells + __umul24(__umul24(__umul24(iz, p.size) + iy, p.size) + ix, kCellF4); // < 2^24 (size <= 65)
#pragma unroll
        for (int i = 0; i < 6; i++) c6[i] = cell[i];
    }
    const float f[24] = {c6[0].x, c6[0].y, c6[0].z, c6[0].w, c6[1].x, c6[1].y, c6[1].z, c6[1].w, c6[2].x, c6[2].y, c6[2].z, c6[2].w,
                         c6[3].x, c6[3].y, c6[3].z, c6[3].w, c6[4].x, c6[4].y, c6[4].z, c6[4].w, c6[5].x, c6[5].y, c6[5].z, c6[5].w};
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = make_float4(f[3 * i], f[3 * i + 1], f[3 * i + 2], 0.0f);
}

// The wave's 3 x 3 x 3 window: anchor = the cell (cx, cy, cz) of the wave's centre pixel minus one per axis, shifted to stay
// inside the table (cell indices run 0 .. size-1; the launchers guarantee size >= 3); three coalesced wave loads into `mine`.
// (A 4 x 4 x 4 window with the three cell indices packed into one word -- one subtraction, one mask test and one v_dot4 for the
// LDS offset instead of nine instructions -- was built on top of the 64 x 16 blocks and measured: 33^3 natural-like 61.6 k -> 52.6 k
// fps, flat bars 24.3 -> 33.6 us per 4K frame, 65^3 unchanged: the three extra wave loads per block and the 6 KB of LDS per wave cost
// more than the simpler test and the wider window return.  A 2 x 2 x 2 window anchored by the centre pixel's position in its cell
// (one wave load): flat bars unchanged, natural-like 61.6 k -> 37.2 k fps -- too many pixels fall outside.)
struct TileRel {
    uint32_t r0, r1, r2; // offsets of this lane's three pieces of the window relative to the anchor cell (float4 units)
};

__device__ __forceinline__ TileRel tile_rel(uint32_t lane, const LutParams &p) // issue early: the values are needed after the coordinates
{
    return {p.tile_tables[2 * kCoordEntries + lane], p.tile_tables[2 * kCoordEntries + 64 + lane], p.tile_tables[2 * kCoordEntries + 128 + lane]};
}

__device__ __forceinline__ void tile_load_window(float4 *mine, uint32_t lane, const LutParams &p, const TileRel &rel, uint32_t cx, uint32_t cy,
                                                 uint32_t cz, uint32_t &ax, uint32_t &ay, uint32_t &az)
{
    const uint32_t rel0 = rel.r0, rel1 = rel.r1, rel2 = rel.r2;
    const uint32_t hi = p.size - 3;
    ax = min(cx > 0 ? cx - 1 : 0u, hi); ay = min(cy > 0 ? cy - 1 : 0u, hi); az = min(cz > 0 ? cz - 1 : 0u, hi);
    const uint32_t anchor = (ax + p.size * (ay + p.size * az)) * kCellF4; // float4 units; wave-uniform
    // piece q = 6 n + i of the window goes to pitch * n + i (lane-constant indices: q / 6 by multiplication, q < 192)
    auto slot = [](uint32_t q) { const uint32_t n = (q * 171u) >> 10; return n * (uint32_t)kTileCellPitch + (q - n * 6u); };
    mine[slot(lane)] = p.cells[anchor + rel0];
    mine[slot(64 + lane)] = p.cells[anchor + rel1];
    if (lane < (uint32_t)kTileNbPieces - 128u) mine[slot(128 + lane)] = p.cells[anchor + rel2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WIDE: RGBA64 (LE: little endian) -- a lane's four pixels are 32 bytes (two 16-byte loads), the lattice coordinates come from
// lf_coord on the 16-bit values (the byte-indexed LDS table does not exist for 65536 values; same arithmetic as the gather
// kernel's lf_px16), the output is lf_px16's.  Round 2: 4K natural-like RGBA64 frame 43.4 us with the per-lane gathers.
// A wave's block: ACROSS lanes x (64 / ACROSS) lanes, every lane ROWS rows of four pixels: 4 ACROSS x (64 / ACROSS) ROWS pixels.
// <16, 4> = 64 x 16 and <8, 2> = 32 x 16 are built (the launcher explains the choice): the coordinate table and the window are set
// up once per 1024 / 512 pixels; the first version's 16 x 16 tile (<4, 1>) paid that set-up every 256 pixels and reached 47.0 k fps
// where 64 x 16 reaches 61.6 k.
template <bool WIDE, bool LE, int ACROSS, int ROWS>
__global__ __launch_bounds__(kBlock) void colorlut_tile_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t height,
                                                               uint32_t in_stride, uint32_t out_stride, LutParams p)
{
    constexpr uint32_t kBpp = WIDE ? 8 : 4;
    // pixels per lane and row: one 16-byte load / store per lane, contiguous over the lanes (RGBA64: two pixels; with four -- two
    // instructions whose lanes sit 32 bytes apart -- a 4K natural-like RGBA64 frame took 32.4 us instead of 30.0 us)
    constexpr int PX = WIDE ? 2 : 4;
    constexpr uint32_t kAcross = ACROSS, kTileW = PX * ACROSS, kDown = 64 / ACROSS, kTileH = kDown * ROWS,
                       kCentreLane = (kDown / 2) * ACROSS + ACROSS / 2;
    __shared__ float4 nbr[kBlock / 64][kTileWaveLdsFloat4];
    __shared__ uint2 coord[WIDE ? 1 : kCoordEntries]; // {cell index, fraction bits} per channel and byte value
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if constexpr (!WIDE) {
        const uint2 *src = reinterpret_cast<const uint2 *>(p.tile_tables);
#pragma unroll
        for (uint32_t i = 0; i < kCoordEntries / kBlock; i++) coord[i * kBlock + threadIdx.x] = src[i * kBlock + threadIdx.x];
    }
    // workgroup = four horizontally adjacent tiles (grid x), one tile row per grid y
    const uint32_t x = (blockIdx.x * (kBlock / 64) + wave) * kTileW + (lane % kAcross) * PX, y0 = blockIdx.y * kTileH + (lane / kAcross) * ROWS;
    const uint8_t *in = in_fb.base[blockIdx.z];
    uint8_t *out = out_fb.base[blockIdx.z];
    const TileRel rel = tile_rel(lane, p);
    uint32_t ax = 0, ay = 0, az = 0;
    const uint32_t wave_lds_bytes = wave * (uint32_t)(kTileWaveLdsFloat4 * sizeof(float4));
#pragma unroll
    for (int row = 0; row < ROWS; row++) {
        const uint32_t y = y0 + row;
        const bool valid = x < width && y < height; // width % 4 == 0 (launcher): a lane's pixels are all inside or all outside
        uint4 v = make_uint4(0, 0, 0, 0);
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        if (valid) {
            if constexpr (WIDE) { // streamed once: non-temporal, the cell table keeps the L2 (as in colorlut_xtile_kernel)
                const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(in + (y * in_stride + x * kBpp)));
                v = make_uint4(t.x, t.y, t.z, t.w);
            } else
                v = *reinterpret_cast<const uint4 *>(in + (y * in_stride + x * kBpp));
        }
        if (row == 0) {
            if constexpr (!WIDE) __syncthreads(); // coordinate table complete
        }
        uint32_t px[4] = {v.x, v.y, v.z, v.w};      // RGBA8: the four pixels; RGBA64: low words (r | g << 16) of the two pixels
        uint32_t px_hi[4] = {0, 0, 0, 0};           // RGBA64: high words (b | a << 16)
        if constexpr (WIDE) { px[0] = v.x; px_hi[0] = v.y; px[1] = v.z; px_hi[1] = v.w; }
        uint32_t ix[4], iy[4], iz[4];
        float fx[4], fy[4], fz[4];
#pragma unroll
        for (int j = 0; j < PX; j++) {
            if constexpr (WIDE) {
                uint32_t rv = px[j] & 0xffffu, gv = px[j] >> 16, bv = px_hi[j] & 0xffffu;
                if constexpr (!LE) { rv = bswap16(rv); gv = bswap16(gv); bv = bswap16(bv); }
                lf_coord((float)rv, p.fast, p.scale[0], p.offset[0], p.size_m1, ix[j], fx[j]);
                lf_coord((float)gv, p.fast, p.scale[1], p.offset[1], p.size_m1, iy[j], fy[j]);
                lf_coord((float)bv, p.fast, p.scale[2], p.offset[2], p.size_m1, iz[j], fz[j]);
            } else {
                const uint2 er = coord[px[j] & 0xffu], eg = coord[256 + ((px[j] >> 8) & 0xffu)], eb = coord[512 + ((px[j] >> 16) & 0xffu)];
                ix[j] = er.x; fx[j] = __uint_as_float(er.y);
                iy[j] = eg.x; fy[j] = __uint_as_float(eg.y);
                iz[j] = eb.x; fz[j] = __uint_as_float(eb.y);
            }
        }
        if (row == 0) {
            // anchor: the cell of the block's centre pixel (64 x 16: lane 40 = rows 8..11, columns 32..35, its first row)
            // (a block that sticks out of the frame on the right or at the bottom may have its centre outside: lane 0 then)
            const uint32_t centre = __builtin_amdgcn_readlane((int)valid, kCentreLane) ? kCentreLane : 0u;
            const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)ix[0], centre),
                           cy = (uint32_t)__builtin_amdgcn_readlane((int)iy[0], centre),
                           cz = (uint32_t)__builtin_amdgcn_readlane((int)iz[0], centre);
            tile_load_window(nbr[wave], lane, p, rel, cx, cy, cz, ax, ay, az);
        }
#pragma unroll
        for (int j = 0; j < PX; j++) {
            float4 c[8];
            tile_cell((lds_bytes_t)&nbr[0][0], wave_lds_bytes, p, ix[j], iy[j], iz[j], ax, ay, az, c);
            float r, g, b;
            lf_trilinear<true>(c, fx[j], fy[j], fz[j], r, g, b);
            // RGBA64: float_to_u16 as one fused multiply-add, trunc(fma(v, 65535, 0.5)) == round(v * 65535) for every float v in [0, 1]
            // (tools/prove_exact.c P15; with pred(0.5) two floats fail for 65535, with 0.5 none); RGBA8 keeps mul + add (P10) here --
            // this kernel is the round-2 reference of the A/B runs
            const float yr = WIDE ? __builtin_fmaf(r, p.fast.out_scale, 0.5f) : r * p.fast.out_scale + p.fast.pred_half,
                        yg = WIDE ? __builtin_fmaf(g, p.fast.out_scale, 0.5f) : g * p.fast.out_scale + p.fast.pred_half,
                        yb = WIDE ? __builtin_fmaf(b, p.fast.out_scale, 0.5f) : b * p.fast.out_scale + p.fast.pred_half;
            if constexpr (WIDE) {
                uint32_t ro = (uint32_t)__float2uint_rz(yr), go = (uint32_t)__float2uint_rz(yg), bo = (uint32_t)__float2uint_rz(yb);
                if constexpr (!LE) { ro = bswap16(ro); go = bswap16(go); bo = bswap16(bo); }
                px[j] = ro | (go << 16);
                px_hi[j] = bo | (px_hi[j] & 0xffff0000u);
            } else {
                uint32_t w = px[j];
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yr));
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yg));
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yb));
                px[j] = w;
            }
        }
        if (valid) {
            if constexpr (WIDE) {
                const u32x4_t t = {px[0], px_hi[0], px[1], px_hi[1]};
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t *>(out + (y * out_stride + x * kBpp)));
            } else {
                *reinterpret_cast<uint4 *>(out + (y * out_stride + x * 4)) = make_uint4(px[0], px[1], px[2], px[3]);
            }
        }
    }
}

// ---------------------------------------------------------------- the x-prelerped tile kernel (round 3)
//
// An RGBA8 pixel's r byte fixes (x0, tx), so the four x-lerps of sample_3d (imp.rs:515-518) depend on (r byte, y node, z node) only:
//   X[y][z][r] = c(x0,y,z) + (c(x1,y,z) - c(x0,y,z)) * tx          (the reference's own three roundings, done once per LUT)
// and the difference the y-lerp subtracts, D[y][z][r] = RN(X[min(y+1,max)][z][r] - X[y][z][r]), is fixed with it.  Per pixel that
// leaves  c0 = X[y0][z0][r] + D[y0][z0][r] * ty,  c1 = X[y0][z1][r] + D[y0][z1][r] * ty,  out = c0 + (c1 - c0) * tz -- 21 f32
// operations instead of 51, two 24-byte LDS reads instead of six 16-byte ones, same bits (every operation that remains is one the
// reference performs, on the same operands).  Table: [y][z][r] with z running to size inclusive -- row `size` repeats row size - 1,
// which is what z1 = min(z0 + 1, max) selects there, so the second entry is ALWAYS the next z row -- 256 x size x (size + 1) entries
// of 24 bytes (33^3: 6.9 MB), built on the device by colorlut_xtable_build_kernel from the node layout and the r channel's
// coordinate table.
// A wave owns a 64 x 20 block of pixels and keeps in wave-private LDS the entries of RW consecutive r bytes x 3 y cells x 4 z rows
// around a mean colour of the block (18 r bytes since round 5: see MVFX_XTILE_RW): 12 rows of RW x 24 contiguous bytes; pixels outside the window read
// their two entries from the table in global memory.  The per-byte coordinate entries of g and b hold the cell index already
// multiplied by the window's LDS pitch of that axis, so the in-window test and the LDS address are three subtractions, three
// compares, one add3 and one mad.
constexpr uint32_t kXRowPieces = 384;    // 16-byte pieces per (y, z) row of the table: 256 entries x 24 B
#ifndef MVFX_XTILE_YPAD
#define MVFX_XTILE_YPAD 0 // bytes between the y slabs of a window in LDS.  Without it a slab is 4 x 576 = 2304 bytes = 9 x 256: the entries
                          // (r, y, z) and (r, y + 1, z) sit on the SAME banks, and lanes of one ds_read whose g bytes fall into neighbouring
                          // cells -- every block of a noisy picture -- serialise
#endif
#ifndef MVFX_XTILE_NY
#define MVFX_XTILE_NY 3 // y cells of a window
#endif
#ifndef MVFX_XTILE_NZ
#define MVFX_XTILE_NZ 3 // z cells of a window (NZ + 1 z rows: a pixel reads rows z0 and z0 + 1)
#endif
constexpr uint32_t kXNY = MVFX_XTILE_NY, kXNZ = MVFX_XTILE_NZ, kXNZR = kXNZ + 1;
constexpr uint32_t kXPitchZ = MVFX_XTILE_RW * 24, kXPitchY = kXNZR * kXPitchZ + MVFX_XTILE_YPAD; // LDS bytes between z rows / y cells of a window
// LDS of a wave's window.  The workgroup's total (4 windows + the 4 KB coordinate table) must stay within 32000 bytes: LDS is handed out
// in granules of 1280 bytes and five workgroups per CU need 5 x 25 granules = 160000 <= 163840; one granule more per workgroup costs a
// workgroup per CU (measured: -6 % on every content).  Unpadded: 6912 + 32 spare bytes; padded: no pad behind the last slab, no spare.
constexpr uint32_t kXWaveBytes = MVFX_XTILE_YPAD ? kXNY * kXPitchY - MVFX_XTILE_YPAD : kXNY * kXPitchY + 32;
static_assert(4 * kXWaveBytes + 4096 <= 32000, "at least five workgroups per CU (six with the shipped 18 r bytes: 24 960 bytes)");
// the window's first cell along an axis of NCELLS cells for an anchor at lattice coordinate `c` (cell + fraction): the anchor's cell in
// the middle (odd), or -- even -- the half of its cell the anchor lies in decides which side gets the extra cell
template <uint32_t NCELLS>
__device__ __forceinline__ uint32_t xtile_first_cell(float c, uint32_t size)
{
    const uint32_t cell = min((uint32_t)c, size - 1);
    const uint32_t below = (NCELLS & 1u) ? (NCELLS - 1u) / 2u : NCELLS / 2u - ((c - (float)cell) >= 0.5f ? 1u : 0u);
    return min(cell > below ? cell - below : 0u, size - NCELLS);
}

// The wave's window: 3 y slabs x 4 z rows x RW entries of the x table, global -> LDS directly (global_load_lds_dwordx4: LDS address =
// wave-uniform base + lane x 16, which is the window's piece order inside a slab): no staging registers, no ds_write pass.
// `base` = the table piece of (ay, az, ar), wave-uniform.
__device__ __forceinline__ void xtile_fill_window(const float4 *xtable, uint32_t base, uint32_t size, uint8_t *lds_region, uint32_t lane)
{
    typedef __attribute__((address_space(3))) void *lds_void_t;
    typedef const __attribute__((address_space(1))) void *global_void_t;
    constexpr uint32_t kRowP = MVFX_XTILE_RW * 3 / 2, kSlabP = kXNZR * kRowP;
    if constexpr (MVFX_XTILE_YPAD == 0) {
        constexpr uint32_t kPieces = kXNY * kSlabP;
#pragma unroll
        for (uint32_t q0 = 0; q0 < kPieces; q0 += 64) {
            const uint32_t q = q0 + lane;
            if (q0 + 64 <= kPieces || q < kPieces) {
                const uint32_t wr = q / kRowP, k = q - wr * kRowP; // window row = dy * (NZ + 1) + dz
                __builtin_amdgcn_global_load_lds((global_void_t)(xtable + (base + ((wr / kXNZR) * (size + 1) + (wr % kXNZR)) * kXRowPieces + k)),
                                                 (lds_void_t)(lds_region + q0 * 16), 16, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (uint32_t dy = 0; dy < kXNY; dy++) {
#pragma unroll
            for (uint32_t q0 = 0; q0 < kSlabP; q0 += 64) {
                const uint32_t q = q0 + lane;
                if (q0 + 64 <= kSlabP || q < kSlabP) {
                    const uint32_t dz = q / kRowP, k = q - dz * kRowP;
                    __builtin_amdgcn_global_load_lds((global_void_t)(xtable + (base + (dy * (size + 1) + dz) * kXRowPieces + k)),
                                                     (lds_void_t)(lds_region + dy * kXPitchY + q0 * 16), 16, 0, 0);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void colorlut_xtable_build_kernel(const float4 *__restrict__ cube, const uint32_t *__restrict__ tile_tables,
                                                                    uint32_t size, float *__restrict__ xtable)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x; // = (y * (size + 1) + zrow) * 256 + r
    if (i >= size * (size + 1) * 256u) return;
    const uint32_t r = i & 255u, yz = i >> 8, zrow = yz % (size + 1), y = yz / (size + 1), m = size - 1, s2 = size * size;
    const uint32_t z = min(zrow, m);
    const uint32_t x0 = tile_tables[2 * r], x1 = min(x0 + 1, m), y1 = min(y + 1, m);
    const float tx = __uint_as_float(tile_tables[2 * r + 1]);
    const float4 a0 = cube[x0 + y * size + z * s2], b0 = cube[x1 + y * size + z * s2];
    const float4 a1 = cube[x0 + y1 * size + z * s2], b1 = cube[x1 + y1 * size + z * s2];
    const float X0[3] = {lf_lerp(a0.x, b0.x, tx), lf_lerp(a0.y, b0.y, tx), lf_lerp(a0.z, b0.z, tx)};
    const float X1[3] = {lf_lerp(a1.x, b1.x, tx), lf_lerp(a1.y, b1.y, tx), lf_lerp(a1.z, b1.z, tx)};
    float *e = xtable + (uint64_t)i * 6;
    e[0] = X0[0]; e[1] = X0[1]; e[2] = X0[2];
    e[3] = X1[0] - X0[0]; e[4] = X1[1] - X0[1]; e[5] = X1[2] - X0[2];
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifndef MVFX_XTILE_READ2
#define MVFX_XTILE_READ2 0 // 1: let the compiler pair the 8-byte window reads into ds_read2_b64 (round 3)
#endif
#if MVFX_XTILE_READ2
typedef const __attribute__((address_space(3))) f32x2_t *lds_float2_t;
#else
// volatile: the six 8-byte reads of a pixel stay six ds_read_b64.  Left alone the compiler pairs them into three ds_read2_b64, which the
// LDS serves at HALF the rate (8 array cycles for 16 bytes per lane, 16-lane groups on 32 banks, against 2 x 2 cycles, 32-lane groups on
// 64 banks: MI355X_MICROARCH.md, LDS table)
typedef const volatile __attribute__((address_space(3))) f32x2_t *lds_float2_t;
#endif

#ifndef MVFX_XTILE_ROWS
#define MVFX_XTILE_ROWS 5 // rows of four pixels per lane: the wave's block is 64 x (4 x rows) pixels.  Round 4, with the register anchor, 16 x 4K natural-like
                          // frames: 8 / 12 / 16 / 20 / 24 rows of pixels 60.5 / 68.9 / 74.4 / 75.5 / 73.6 k fps; 20 against 16: +-3 +3 %, +-8 +2 %, +-16 -2 %,
                          // flat bars 63.2 -> 65.6 k, 94 VGPRs (16: 90), one frame per launch unchanged
#endif
#ifndef MVFX_XTILE_SAMPLE_ROW
#define MVFX_XTILE_SAMPLE_ROW 1 // ANCHOR4 == 7: which of the lane's rows the sample comes from
#endif
#ifndef MVFX_I420_XTILE_MEAN
#define MVFX_I420_XTILE_MEAN 1 // colorlut_i420_xtile_kernel: window anchored at the mean of the lanes' first pixels (0: lane 36's)
#endif
#ifndef MVFX_XTILE_SPREAD64
#define MVFX_XTILE_SPREAD64 120 // ANCHOR4 == 7: corner samples (60 x 12 pixels apart) differ by more than this along both diagonals -> an edge
#endif
#ifndef MVFX_XTILE_SPREAD_LOW
#define MVFX_XTILE_SPREAD_LOW 20 // ANCHOR4 == 7: four lanes around the centre agree this closely (sum of absolute byte differences along both diagonals) -> their mean
#endif
#ifndef MVFX_XTILE_ANCHOR4
#define MVFX_XTILE_ANCHOR4 7 // where the window is anchored.  0: at the block's centre pixel (rounds 3 and 4 until its last day): one pixel carries the
                             // full noise of the picture, and every code the anchor is off shrinks the part of the window the other pixels can use.
                             // 1: at the mean of four pixels of the block (the centres of its quadrants) through scalar loads.
                             // 2: the same four samples through ONE vector load of lanes 0..3 ahead of the pixel loads, the mean only
                             // where they agree (across an edge the mean fits neither side: the first sample stands).
                             // Round 4, after the window reads became ds_read_b64 (profiles/r4/colorlut_anchor.txt, same box, 16 x 4K per launch):
                             //   noise +-0 / 3 / 5 / 8 / 16 / flat bars:  0: 76.4 / 73.2 / 59.9 / 39.2 / 23.9 / 64.5 k fps
                             //                                            1: 73.7 / 72.0 / 66.7 / 44.9 / 25.4 / 61.9 k
                             //                                            2: 76.6 / 74.0 / 68.1 / 46.3 / 25.0 / 63.7 k   one frame per launch 18.5 us (0: 19.0)
                             // (round 3, with ds_read2_b64 window reads, 2 cost the clean frames 1-2 % and a single frame 1 us: it stayed off)
                             // 7 (shipped): samples out of the pixel registers, no load of their own -- the mean of four lanes around the centre where
                             // they agree, of all sixty-four lanes elsewhere (see the kernel).  Same box as a run of 2:
                             //   noise +-0 / 3 / 5 / 8 / 16:  2: 73.7 / 70.8 / 64.6 / 46.6 / 25.1 k fps     7: 72.9 / 70.4 / 64.8 / 52.0 / 26.9 k
                             //   (sixteen samples through the vector load of 2: 69.7 / 68.1 / 65.0 / 49.2 / 25.7 k -- the load's own lines cost more
                             //   than the better mean returns on clean frames)
#endif
#ifndef MVFX_XTILE_MIN_BLOCKS
#define MVFX_XTILE_MIN_BLOCKS 1
#endif
#ifndef MVFX_XTILE_FAR_GATHER
#define MVFX_XTILE_FAR_GATHER 1 // 1 (round 4): blocks of far-apart colours skip the window (see the kernel)
#endif
#ifndef MVFX_XTILE_FAR
#define MVFX_XTILE_FAR 64 // |g - centre g| + |b - centre b| above which an outside pixel counts as far
#endif
#ifndef MVFX_XTILE_NT
#define MVFX_XTILE_NT 1   // 1: non-temporal pixel loads and stores (16 x 4K natural-like 70.7 k -> 73.1 k fps, one frame 21.8 -> 19.1 us:
                          // the pixels stream through once, the table stays in L2)
#endif
template <int RW>
__global__ __launch_bounds__(kBlock, MVFX_XTILE_MIN_BLOCKS) void colorlut_xtile_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t height,
                                                                uint32_t in_stride, uint32_t out_stride, LutParams p)
{
    static_assert(RW % 2 == 0 && RW <= 64, "window rows start and end on 16-byte pieces");
    static_assert(RW * 24 == kXPitchZ, "the coordinate table is built for this window width");
    constexpr uint32_t kAcross = 16, kRows = MVFX_XTILE_ROWS, kTileW = 64, kTileH = 4 * kRows;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    constexpr uint32_t kWaveBytes = kXWaveBytes;
    __shared__ __attribute__((aligned(16))) uint8_t win[(kBlock / 64) * kWaveBytes];
    __shared__ uint2 coord[512]; // {cell index x LDS pitch, fraction bits} per byte value of the g and b channels
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (giving every XCD a contiguous run of the workgroup order -- whole frames of a batch, a band of a single frame -- so that its L2
    // holds a smaller part of the table: 71.6 k vs 72.3 k fps, one frame 20.8 vs 19.1 us; with non-temporal pixel accesses the table
    // misses are 8 % of the pixel bytes already, FETCH_SIZE 572 MB vs 540 MB per 16 frames)
    const uint32_t gx = blockIdx.x, gy = blockIdx.y, gz = blockIdx.z;
    const uint8_t *in = in_fb.base[gz];
    uint8_t *out = out_fb.base[gz];
    const uint32_t bx = (gx * (kBlock / 64) + wave) * kTileW, by = gy * kTileH; // the wave's block
    const uint32_t x = bx + (lane % kAcross) * 4, y0 = by + (lane / kAcross) * kRows;
#if MVFX_XTILE_ANCHOR4 == 2
    // four samples of the block (the centres of its quadrants) through ONE vector load of lanes 0..3, issued ahead of the pixel loads
    uint32_t smp = 0;
    const bool whole_block = bx + kTileW <= width && by + kTileH <= height;
    if (whole_block && lane < 4)
        smp = *reinterpret_cast<const uint32_t *>(in + ((by + kTileH / 4 + (lane >> 1) * (kTileH / 2)) * in_stride + (bx + kTileW / 4 + (lane & 1) * (kTileW / 2)) * 4));
#elif MVFX_XTILE_ANCHOR4 == 7
    const bool whole_block = bx + kTileW <= width && by + kTileH <= height;
#endif
    // 1. every pixel of the lane, up front (four 16-byte loads in flight while the window is being fetched)
    uint32_t voff_in = y0 * in_stride + x * 4, voff_out = y0 * out_stride + x * 4; // the lane's byte offsets into rows y0 .. of the frames
    asm volatile("" : "+v"(voff_in), "+v"(voff_out)); // both formed HERE (left alone the compiler re-forms the second one late from a 64-bit x * 4 that it spills)
    uint4 v[kRows];
#pragma unroll
    for (uint32_t row = 0; row < kRows; row++) {
        v[row] = make_uint4(0, 0, 0, 0);
        if (x < width && y0 + row < height) {
            const u32x4_t *src = reinterpret_cast<const u32x4_t *>(in + (size_t)row * in_stride + voff_in); // uniform row base + one 32-bit lane offset
            const u32x4_t t = MVFX_XTILE_NT ? __builtin_nontemporal_load(src) : *src;
            v[row] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
    // 2. the window, anchored at the block's centre pixel (its top-left pixel when the centre lies outside the frame): the pixel and
    // its two coordinate entries come through the scalar cache, so this chain does not wait for the vector loads above
    uint32_t ar, ayp, azp; // anchor: first r byte, y cell x kXPitchY, z row x kXPitchZ
    uint32_t ccpx; // the block's centre pixel (scalar)
    {
        // (a wave of the last workgroup of a row may lie wholly right of the frame: it reads pixel (0, 0) and stores nothing)
        const uint32_t cxp = bx + kTileW / 2 < width ? bx + kTileW / 2 : bx, cyp = by + kTileH / 2 < height ? by + kTileH / 2 : by;
        const uint32_t coff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bx < width ? cyp * in_stride + cxp * 4 : 0u));
        uint32_t cpx = *reinterpret_cast<const uint32_t *>(in + coff);
#if MVFX_XTILE_ANCHOR4 == 2
        if (__builtin_amdgcn_readfirstlane((int)whole_block)) {
            const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)smp, 0) & 0xffffffu, q1 = (uint32_t)__builtin_amdgcn_readlane((int)smp, 1) & 0xffffffu,
                           q2 = (uint32_t)__builtin_amdgcn_readlane((int)smp, 2) & 0xffffffu, q3 = (uint32_t)__builtin_amdgcn_readlane((int)smp, 3) & 0xffffffu;
            // the mean only where the four agree (sum of absolute byte differences along the two diagonals): across an edge the mean
            // would fit neither side, there the first sample stands
            const uint32_t spread = __builtin_amdgcn_sad_u8(q0, q3, 0u) + __builtin_amdgcn_sad_u8(q1, q2, 0u);
            const uint32_t ev = (q0 & 0x00ff00ffu) + (q1 & 0x00ff00ffu) + (q2 & 0x00ff00ffu) + (q3 & 0x00ff00ffu) + 0x00020002u;
            const uint32_t od = ((q0 >> 8) & 0x00ff00ffu) + ((q1 >> 8) & 0x00ff00ffu) + ((q2 >> 8) & 0x00ff00ffu) + ((q3 >> 8) & 0x00ff00ffu) + 0x00020002u;
            const uint32_t mean = ((ev >> 2) & 0x00ff00ffu) | (((od >> 2) & 0x00ff00ffu) << 8);
            cpx = spread <= 72u ? mean : q0;
        }
        cpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cpx);
#elif MVFX_XTILE_ANCHOR4 == 7
        // Samples that cost no memory access at all: every lane's own pixel (x + 1, y0 + 1), out of the registers the pixel loads above
        // fill -- a 16 x 4 lattice over the block.  (A separate sample load fetches lines of its own: with sixteen lanes taking part in it
        // the clean frames lost 4-5 %, profiles/r4/colorlut_anchor.txt.)  The window fill now waits for the second row's pixel load
        // instead of the sample load: the same memory round trip.  Where four lanes around the block's centre agree closely (clean
        // content) their mean is the anchor; elsewhere (noise, texture) the mean of all sixty-four -- eight dependent DPP additions on the
        // path the window fill waits for, which clean blocks do not pay.
        // sixty-four samples that cost no memory access at all: every lane's own pixel (x + 1, y0 + 1), out of the registers the pixel loads
        // above fill -- a 16 x 4 lattice over the block.  (A separate sample load fetches lines of its own: with sixteen lanes taking part
        // in it the clean frames lost 4-5 %, profiles/r4/colorlut_anchor.txt.)  The window fill now waits for the second row's pixel load
        // instead of the sample load: the same memory round trip.
        if (__builtin_amdgcn_readfirstlane((int)whole_block)) {
            const uint32_t mine = v[MVFX_XTILE_SAMPLE_ROW < kRows ? MVFX_XTILE_SAMPLE_ROW : 0].y;
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 21) & 0xffffffu, i1 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 26) & 0xffffffu,
                           i2 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 37) & 0xffffffu, i3 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 42) & 0xffffffu;
            const uint32_t inner = __builtin_amdgcn_sad_u8(i0, i3, 0u) + __builtin_amdgcn_sad_u8(i1, i2, 0u);
            if (inner <= (uint32_t)MVFX_XTILE_SPREAD_LOW) {
                const uint32_t ev4 = (i0 & 0x00ff00ffu) + (i1 & 0x00ff00ffu) + (i2 & 0x00ff00ffu) + (i3 & 0x00ff00ffu) + 0x00020002u;
                const uint32_t od4 = ((i0 >> 8) & 0x00ff00ffu) + ((i1 >> 8) & 0x00ff00ffu) + ((i2 >> 8) & 0x00ff00ffu) + ((i3 >> 8) & 0x00ff00ffu) + 0x00020002u;
                cpx = ((ev4 >> 2) & 0x00ff00ffu) | (((od4 >> 2) & 0x000000ffu) << 8);
            } else {
            uint32_t ev = mine & 0x00ff00ffu, od = (mine >> 8) & 0x00ff00ffu;
#define MVFX_ROW_ADD(v_, ctrl) v_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v_, ctrl, 0xf, 0xf, true)
            MVFX_ROW_ADD(ev, 0x111); MVFX_ROW_ADD(od, 0x111);
            MVFX_ROW_ADD(ev, 0x112); MVFX_ROW_ADD(od, 0x112);
            MVFX_ROW_ADD(ev, 0x114); MVFX_ROW_ADD(od, 0x114);
            MVFX_ROW_ADD(ev, 0x118); MVFX_ROW_ADD(od, 0x118);
#undef MVFX_ROW_ADD
            // lanes 15, 31, 47, 63 hold their row's sums (16 x 255 fits twelve bits; the four rows together fourteen)
            const uint32_t sev = (uint32_t)__builtin_amdgcn_readlane((int)ev, 15) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 31) +
                                 (uint32_t)__builtin_amdgcn_readlane((int)ev, 47) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 63) + 0x00200020u;
            const uint32_t sod = (uint32_t)__builtin_amdgcn_readlane((int)od, 15) + (uint32_t)__builtin_amdgcn_readlane((int)od, 31) +
                                 (uint32_t)__builtin_amdgcn_readlane((int)od, 47) + (uint32_t)__builtin_amdgcn_readlane((int)od, 63) + 0x00200020u;
            const uint32_t mean = ((sev >> 6) & 0x00ff00ffu) | (((sod >> 6) & 0x000000ffu) << 8);
            const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 0) & 0xffffffu, q1 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 15) & 0xffffffu,
                           q2 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 48) & 0xffffffu, q3 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 63) & 0xffffffu;
            const uint32_t spread = __builtin_amdgcn_sad_u8(q0, q3, 0u) + __builtin_amdgcn_sad_u8(q1, q2, 0u);
            // across an edge the mean fits neither side: the lane next to the block's centre stands
            cpx = spread <= (uint32_t)MVFX_XTILE_SPREAD64 ? mean : ((uint32_t)__builtin_amdgcn_readlane((int)mine, 40) & 0xffffffu);
            }
        }
        cpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cpx);
#elif MVFX_XTILE_ANCHOR4
        // a block that lies wholly inside the frame is anchored at the MEAN of four of its pixels (the centres of its quadrants): one pixel
        // carries the full noise of the picture, and every code the anchor is off shrinks the part of the window the other pixels can use
        // (gradients +- 5 codes of noise: profiles/r3/colorlut_anchor4.txt)
        if (bx + kTileW <= width && by + kTileH <= height) {
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((by + kTileH / 4) * in_stride + (bx + kTileW / 4) * 4));
            const uint32_t dxq = (kTileW / 2) * 4, dyq = (kTileH / 2) * in_stride;
            const uint32_t q0 = *reinterpret_cast<const uint32_t *>(in + o0), q1 = *reinterpret_cast<const uint32_t *>(in + o0 + dxq),
                           q2 = *reinterpret_cast<const uint32_t *>(in + o0 + dyq), q3 = *reinterpret_cast<const uint32_t *>(in + o0 + dyq + dxq);
            // per-byte sums of four bytes fit ten bits: even and odd bytes in separate words
            const uint32_t ev = (q0 & 0x00ff00ffu) + (q1 & 0x00ff00ffu) + (q2 & 0x00ff00ffu) + (q3 & 0x00ff00ffu) + 0x00020002u;
            const uint32_t od = ((q0 >> 8) & 0x00ff00ffu) + ((q1 >> 8) & 0x00ff00ffu) + ((q2 >> 8) & 0x00ff00ffu) + ((q3 >> 8) & 0x00ff00ffu) + 0x00020002u;
            cpx = ((ev >> 2) & 0x00ff00ffu) | (((od >> 2) & 0x00ff00ffu) << 8);
        }
        cpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cpx); // wave-uniform by construction: keeps the two table look-ups below scalar
#endif
        const uint32_t cr = cpx & 0xffu;
#ifndef MVFX_XTILE_ANCHOR_ARITH
#define MVFX_XTILE_ANCHOR_ARITH 1 // 1 (round 5): the anchor's y / z cells by arithmetic instead of two scalar loads from the coordinate table
#endif
        // Where the window goes decides how many pixels find their entries in it, never what they compute: the anchor's lattice
        // coordinates may be formed any way.  The two table look-ups of rounds 3 and 4 were scalar loads whose address depends on the
        // pixels -- one more memory round trip on the chain pixel loads -> anchor -> window fill that every wave walks before its first
        // row; the same lattice arithmetic in a handful of VALU operations on the (uniform) anchor colour: +0.7 % calm, +3 % at +-8.
        const float gy = (float)((cpx >> 8) & 0xffu) * (1.0f / 255.0f), bz = (float)((cpx >> 16) & 0xffu) * (1.0f / 255.0f);
        const float ny = fminf(fmaxf(gy * p.scale[1] + p.offset[1], 0.0f), 1.0f) * p.size_m1, nz = fminf(fmaxf(bz * p.scale[2] + p.offset[2], 0.0f), 1.0f) * p.size_m1;
        ccpx = cpx;
        ar = min((cr > RW / 2 ? cr - RW / 2 : 0u) & ~1u, 256u - RW); // even: a window row starts on a 16-byte piece
        const uint32_t ay = (uint32_t)__builtin_amdgcn_readfirstlane((int)xtile_first_cell<kXNY>(ny, p.size)),
                       az = (uint32_t)__builtin_amdgcn_readfirstlane((int)xtile_first_cell<kXNZ>(nz, p.size)); // z rows run 0 .. size
        ayp = ay * kXPitchY;
        azp = az * kXPitchZ;
        const uint32_t base = (ay * (p.size + 1) + az) * kXRowPieces + ar * 3 / 2; // wave-uniform
        xtile_fill_window(p.xtable, base, p.size, win + wave * kWaveBytes, lane);
    }
    coord[threadIdx.x] = p.xcoord[threadIdx.x];
    coord[kBlock + threadIdx.x] = p.xcoord[kBlock + threadIdx.x];
    __syncthreads(); // coordinate table and (a fortiori) this wave's window complete
    // LDS byte address of entry (y cell, z row, r) = yp + zp + 24 r + lds_k, with the anchor folded into the wave-uniform lds_k
    const uint32_t lds_k = wave * kWaveBytes - ayp - azp - ar * 24u, ar24 = ar * 24u, wave_lds = wave * kWaveBytes;
    // One row of four pixels per lane: their entries from the window; a pixel outside it reads the window's first entry and is patched
    // in ONE branch per row (the scalar side of an if / else costs about five instructions) with its two entries of the x table from
    // global memory.
    // Round 4, row 0 only: when nearly all (> 248) of the wave's 256 pixels of that row are outside the window (videotestsrc's grey snow
    // -- r = g = b, uniform-random -- leaves 233 +- 5 outside and is served faster by the branch: its entries all lie on the table's
    // diagonal) AND most of those are far
    // from the block's centre colour (more than MVFX_XTILE_FAR codes in g and b together: not a noisy picture) AND less than a quarter
    // of them resemble the first one (not the other side of an edge between two flat colours -- there the branch serves whole groups of
    // lanes from the same few lines): uniform-random colours.  Such a block is not worth a
    // window at all: every lane gathers its own pixels' 96-byte cells of the cell-packed table (3.45 MB for 33^3: it stays in the XCD's
    // L2, where the 6.9 MB x table does not) the way colorlut_fast_global_kernel does -- lf_px8, the full trilinear form, a cell the lane
    // used last is kept.  Uniform-random 4K frames: 7.2 k -> 12 k fps, HBM traffic 14.5 x -> see profiles/r4.
    // (Also built and measured in round 4, bit-exact, not shipped: listing the outside pixels per wave (ballot + mbcnt) in the LDS of the
    // dead window and serving them densely from a second 6 x 6 x 6 node window: +8 % at +- 8 codes of noise, +7 % at +- 16, -3 % at
    // +- 5, -7 % on flat bars, and 98 VGPRs -- a wave per SIMD less for every block; profiles/r4/colorlut_dense_pass.txt and the
    // commit before this one.)
#ifndef MVFX_XTILE_PP
#define MVFX_XTILE_PP 2 // pixels per pass of a row.  2 (round 5): a row of four pixels in two passes of two -- 56-60 VGPRs instead of 94, which with the
                        // narrower window (MVFX_XTILE_RW) puts six waves on a SIMD.  This version has no "far" test (a block of uniform-random
                        // colours): pictures like that are the other kernel's (the content probe), a stray block is served pixel by pixel.
                        // 4: rounds 3 and 4 (kept for A/B builds).
#endif
#if MVFX_XTILE_PP == 2
    auto do_row = [&](const uint32_t row) -> bool {
        const bool valid = x < width && y0 + row < height;
        uint32_t px[4] = {v[row].x, v[row].y, v[row].z, v[row].w};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f32x2_t e0[2][3], e1[2][3];
            float ty[2], tz[2];
            bool miss[2];
            bool any_miss = false;
#ifndef MVFX_XTILE_PP2_COORD_FIRST
#define MVFX_XTILE_PP2_COORD_FIRST 1 // the four coordinate reads of a pass before its entry reads (two LDS round trips per pass instead of three: +1 % on calm
                                     // frames; with four-pixel passes the same idea cost a wave per SIMD and 6 %)
#endif
            uint2 egs[2], ebs[2];
            if (MVFX_XTILE_PP2_COORD_FIRST) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    egs[jj] = coord[(px[2 * h + jj] >> 8) & 0xffu];
                    ebs[jj] = coord[256 + ((px[2 * h + jj] >> 16) & 0xffu)];
                }
#pragma unroll
                for (int jj = 0; jj < 2; jj++) asm volatile("" : "+v"(egs[jj]), "+v"(ebs[jj]));
            }
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const uint32_t pxj = px[2 * h + jj];
                const uint2 eg = MVFX_XTILE_PP2_COORD_FIRST ? egs[jj] : coord[(pxj >> 8) & 0xffu], eb = MVFX_XTILE_PP2_COORD_FIRST ? ebs[jj] : coord[256 + ((pxj >> 16) & 0xffu)];
                ty[jj] = __uint_as_float(eg.y);
                tz[jj] = __uint_as_float(eb.y);
                uint32_t r24;
                asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r24) : "v"(pxj), "v"(24u));
                const uint32_t dr24 = r24 - ar24, dyp = eg.x - ayp, dzp = eb.x - azp;
                miss[jj] = (dr24 >= (uint32_t)RW * 24u) | (dyp >= kXNY * kXPitchY) | (dzp >= kXNZ * kXPitchZ);
                any_miss = any_miss | miss[jj];
                const uint32_t off = miss[jj] ? wave_lds : eg.x + eb.x + (r24 + lds_k);
                const lds_float2_t q0 = (lds_float2_t)((lds_bytes_t)&win[0] + off), q1 = (lds_float2_t)((lds_bytes_t)&win[0] + off + kXPitchZ);
                e0[jj][0] = q0[0]; e0[jj][1] = q0[1]; e0[jj][2] = q0[2];
                e1[jj][0] = q1[0]; e1[jj][1] = q1[1]; e1[jj][2] = q1[2];
            }
            if (any_miss) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    if (miss[jj]) {
                        const uint32_t pxj = px[2 * h + jj];
                        const uint32_t iy = coord[(pxj >> 8) & 0xffu].x / kXPitchY, iz = coord[256 + ((pxj >> 16) & 0xffu)].x / kXPitchZ, r = pxj & 0xffu;
                        const f32x2_t *g0p = reinterpret_cast<const f32x2_t *>(p.xtable) + (uint64_t)((iy * (p.size + 1) + iz) * 256u + r) * 3, *g1p = g0p + 256 * 3;
                        e0[jj][0] = g0p[0]; e0[jj][1] = g0p[1]; e0[jj][2] = g0p[2];
                        e1[jj][0] = g1p[0]; e1[jj][1] = g1p[1]; e1[jj][2] = g1p[2];
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const float c0r = e0[jj][0].x + e0[jj][1].y * ty[jj], c0g = e0[jj][0].y + e0[jj][2].x * ty[jj], c0b = e0[jj][1].x + e0[jj][2].y * ty[jj];
                const float c1r = e1[jj][0].x + e1[jj][1].y * ty[jj], c1g = e1[jj][0].y + e1[jj][2].x * ty[jj], c1b = e1[jj][1].x + e1[jj][2].y * ty[jj];
                const float rr = lf_add_clamp(c0r, (c1r - c0r) * tz[jj]), gg = lf_add_clamp(c0g, (c1g - c0g) * tz[jj]),
                            bb = lf_add_clamp(c0b, (c1b - c0b) * tz[jj]);
                const float yr = __builtin_fmaf(rr, p.fast.out_scale, p.fast.pred_half), yg = __builtin_fmaf(gg, p.fast.out_scale, p.fast.pred_half),
                            yb = __builtin_fmaf(bb, p.fast.out_scale, p.fast.pred_half);
                uint32_t w = px[2 * h + jj];
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yr));
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yg));
                asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yb));
                px[2 * h + jj] = w;
            }
            __builtin_amdgcn_sched_barrier(0); // the two halves stay two passes
        }
        if (valid) {
            u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)row * out_stride + voff_out);
            const u32x4_t t = {px[0], px[1], px[2], px[3]};
            if (MVFX_XTILE_NT) __builtin_nontemporal_store(t, dst);
            else *dst = t;
        }
        return false;
    };
#else
    auto do_row = [&](const uint32_t row) -> bool { // true: row 0 found the block "far" -- nothing served, nothing stored
        const bool valid = x < width && y0 + row < height; // width % 4 == 0 (launcher)
        uint32_t px[4] = {v[row].x, v[row].y, v[row].z, v[row].w};
        f32x2_t e0[4][3], e1[4][3];
        float ty[4], tz[4];
        bool miss[4];
        bool any_miss = false;
        uint32_t outside = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t pxj = px[j];
            const uint2 eg = coord[(pxj >> 8) & 0xffu], eb = coord[256 + ((pxj >> 16) & 0xffu)];
            ty[j] = __uint_as_float(eg.y);
            tz[j] = __uint_as_float(eb.y);
            uint32_t r24;
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r24) : "v"(pxj), "v"(24u));
            const uint32_t dr24 = r24 - ar24, dyp = eg.x - ayp, dzp = eb.x - azp; // unsigned: below the anchor wraps to a huge value
            // (bitwise |: with || the compiler turns the second and third test into branches behind the LDS wait)
            miss[j] = (dr24 >= (uint32_t)RW * 24u) | (dyp >= kXNY * kXPitchY) | (dzp >= kXNZ * kXPitchZ);
            any_miss = any_miss | miss[j];
            if (MVFX_XTILE_FAR_GATHER && row == 0) outside += (uint32_t)__popcll(__ballot(miss[j] & valid));
            const uint32_t off = miss[j] ? wave_lds : eg.x + eb.x + (r24 + lds_k);
            const lds_float2_t q0 = (lds_float2_t)((lds_bytes_t)&win[0] + off), q1 = (lds_float2_t)((lds_bytes_t)&win[0] + off + kXPitchZ);
            e0[j][0] = q0[0]; e0[j][1] = q0[1]; e0[j][2] = q0[2];
            e1[j][0] = q1[0]; e1[j][1] = q1[1]; e1[j][2] = q1[2];
        }
        if (MVFX_XTILE_FAR_GATHER && row == 0 && outside > 248u) { // wave-uniform, rare
            uint32_t far = 0, alike = 0, fpx = 0;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t b = __ballot(miss[j] & valid);
                if (b != 0 && !found) {
                    fpx = (uint32_t)__builtin_amdgcn_readlane((int)px[j], __builtin_ctzll(b)); // the first outside pixel
                    found = true;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // |g - centre g| + |b - centre b| in one v_sad_u8; "alike": g and b in the first outside pixel's buckets of 16 codes
                far += (uint32_t)__popcll(__ballot(miss[j] & valid & (__builtin_amdgcn_sad_u8(px[j] & 0x00ffff00u, ccpx & 0x00ffff00u, 0u) > (uint32_t)MVFX_XTILE_FAR)));
                alike += (uint32_t)__popcll(__ballot(miss[j] & valid & (((px[j] ^ fpx) & 0x00f0f000u) == 0u)));
            }
            if (far * 4u > outside * 3u && alike * 4u < outside) return true;
        }
        if (any_miss) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (miss[j]) {
                    // (the premultiplied cell indices are read again here rather than kept from above: fewer VGPRs on the path every block takes)
                    const uint32_t pxj = px[j];
                    const uint32_t iy = coord[(pxj >> 8) & 0xffu].x / kXPitchY, iz = coord[256 + ((pxj >> 16) & 0xffu)].x / kXPitchZ, r = pxj & 0xffu;
                    const f32x2_t *g0p = reinterpret_cast<const f32x2_t *>(p.xtable) + (uint64_t)((iy * (p.size + 1) + iz) * 256u + r) * 3, *g1p = g0p + 256 * 3;
                    e0[j][0] = g0p[0]; e0[j][1] = g0p[1]; e0[j][2] = g0p[2];
                    e1[j][0] = g1p[0]; e1[j][1] = g1p[1]; e1[j][2] = g1p[2];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // entry = (X.r, X.g) (X.b, D.r) (D.g, D.b)
            const float c0r = e0[j][0].x + e0[j][1].y * ty[j], c0g = e0[j][0].y + e0[j][2].x * ty[j], c0b = e0[j][1].x + e0[j][2].y * ty[j];
            const float c1r = e1[j][0].x + e1[j][1].y * ty[j], c1g = e1[j][0].y + e1[j][2].x * ty[j], c1b = e1[j][1].x + e1[j][2].y * ty[j];
            const float rr = lf_add_clamp(c0r, (c1r - c0r) * tz[j]), gg = lf_add_clamp(c0g, (c1g - c0g) * tz[j]),
                        bb = lf_add_clamp(c0b, (c1b - c0b) * tz[j]);
            // float_to_u8 (imp.rs:537-539) as ONE fused multiply-add + truncation: trunc(fma(v, 255, pred(0.5))) == round(v * 255) for
            // every float v in [0, 1] (tools/prove_exact.c P15, exhaustive); the other kernels of this file use mul + add (P10)
            const float yr = __builtin_fmaf(rr, p.fast.out_scale, p.fast.pred_half), yg = __builtin_fmaf(gg, p.fast.out_scale, p.fast.pred_half),
                        yb = __builtin_fmaf(bb, p.fast.out_scale, p.fast.pred_half);
            uint32_t w = px[j];
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yr));
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yg));
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yb));
            px[j] = w;
        }
        if (valid) {
            u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)row * out_stride + voff_out);
            const u32x4_t t = {px[0], px[1], px[2], px[3]};
            if (MVFX_XTILE_NT) __builtin_nontemporal_store(t, dst);
            else *dst = t;
        }
        return false;
    };
#endif
    if (do_row(0)) { // wave-uniform, rare
        CellCache cache;
#pragma unroll
        for (uint32_t hr = 0; hr < kRows; hr++) {
            uint4 q = v[hr];
            q.x = lf_px8<true, true>(q.x, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.y = lf_px8<true, true>(q.y, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.z = lf_px8<true, true>(q.z, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.w = lf_px8<true, true>(q.w, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            if (x < width && y0 + hr < height) {
                u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)hr * out_stride + voff_out);
                const u32x4_t t = {q.x, q.y, q.z, q.w};
                if (MVFX_XTILE_NT) __builtin_nontemporal_store(t, dst);
                else *dst = t;
            }
        }
        return;
    }
#pragma unroll
    for (uint32_t row = 1; row < kRows; row++) do_row(row);
}

// ---------------------------------------------------------------- the workgroup-window kernel (round 5)
//
// What round 4's per-wave windows cost, measured by leaving parts of colorlut_xtile_kernel out (profiles/r5/colorlut_experiments.txt, 16 x 4K
// natural-like frames per launch): with the pixels outside the window simply left wrong the kernel runs 72-75 k fps at EVERY noise level --
// the miss service is the whole price of noisy content (+-8 codes: 2.1-2.9 % of the pixels outside a window of 24 r bytes x 3 x 3 cells, but
// 55 % of a wave's (row, j) passes have one; +-16: 61 %), the LDS conflicts of scattered colours cost 10 %.  Serving the misses later, in
// one dense pass per wave (two round trips instead of eleven), bought +5 % at +-8 and nothing at +-16; four blocks per wave with the next
// block's pixels prefetched and the window kept where the anchor stays put bought nothing either (the patch of both: profiles/r5/).  A
// bigger window per wave costs occupancy faster than it saves misses (24 x 4 x 4: 63 k fps on clean frames against 77 k).
// The four waves of a workgroup keep four near-identical windows.  Here they keep ONE: the workgroup owns a 128 x 40 block of pixels (2 x 2
// waves of 64 x 20), the window is 38 r bytes x 5 y cells x 5 z cells (6 z rows) = 27 360 bytes -- the LDS of four 24 x 3 x 3 windows --
// anchored at the mean of the four waves' means.  CPU model of the hit rate on the bench's frames (tools/sim/colorlut_shared_sim.py):
// outside pixels at +-8 codes of noise 2.1 % -> 0.0 %, at +-16 codes 61 % -> 4 %.  Same entries, same arithmetic as colorlut_xtile_kernel:
// same bits.  The rare outside pixel is served in its row pass from the x table in global memory, as in rounds 3 and 4.
#ifndef MVFX_XWG_RW
#define MVFX_XWG_RW 38 // r bytes of the workgroup's window
#endif
#ifndef MVFX_XWG_N
#define MVFX_XWG_N 5   // y and z cells of the workgroup's window
#endif
#ifndef MVFX_XWG_NY
#define MVFX_XWG_NY MVFX_XWG_N
#endif
#ifndef MVFX_XWG_NZ
#define MVFX_XWG_NZ MVFX_XWG_N
#endif
constexpr uint32_t kWgRW = MVFX_XWG_RW, kWgNY = MVFX_XWG_NY, kWgNZ = MVFX_XWG_NZ, kWgNZR = kWgNZ + 1;
constexpr uint32_t kWgPitchZ = kWgRW * 24, kWgPitchY = kWgNZR * kWgPitchZ, kWgWinBytes = kWgNY * kWgPitchY;
static_assert(kWgRW % 2 == 0, "window rows start and end on 16-byte pieces");
static_assert(kWgWinBytes + 4096 + 16 <= 32000, "five workgroups per CU (LDS comes in granules of 1280 bytes: 25 per workgroup)");

__global__ __launch_bounds__(kBlock) void colorlut_xwg_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t height, uint32_t in_stride,
                                                             uint32_t out_stride, LutParams p)
{
    constexpr uint32_t kAcross = 16, kRows = MVFX_XTILE_ROWS, kTileW = 64, kTileH = 4 * kRows, RW = kWgRW;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint8_t win[kWgWinBytes];
    __shared__ uint2 coord[512]; // {cell index x LDS pitch, fraction bits} per byte value of the g and b channels (this kernel's pitches)
    __shared__ uint32_t wave_anchor[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gx = blockIdx.x, gy = blockIdx.y, gz = blockIdx.z;
    const uint8_t *in = in_fb.base[gz];
    uint8_t *out = out_fb.base[gz];
    const uint32_t bx = (gx * 2 + (wave & 1u)) * kTileW, by = (gy * 2 + (wave >> 1)) * kTileH; // the wave's block
    const uint32_t x = bx + (lane % kAcross) * 4, y0 = by + (lane / kAcross) * kRows;
    // 1. every pixel of the lane, up front
    uint32_t voff_in = y0 * in_stride + x * 4, voff_out = y0 * out_stride + x * 4; // the lane's byte offsets into rows y0 .. of the frames
    asm volatile("" : "+v"(voff_in), "+v"(voff_out)); // both formed HERE (colorlut_xtile_kernel)
    uint4 v[kRows];
#pragma unroll
    for (uint32_t row = 0; row < kRows; row++) {
        v[row] = make_uint4(0, 0, 0, 0);
        if (x < width && y0 + row < height) {
            const u32x4_t *src = reinterpret_cast<const u32x4_t *>(in + (size_t)row * in_stride + voff_in);
            const u32x4_t t = MVFX_XTILE_NT ? __builtin_nontemporal_load(src) : *src;
            v[row] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
    coord[threadIdx.x] = p.xcoord_wg[threadIdx.x];
    coord[kBlock + threadIdx.x] = p.xcoord_wg[kBlock + threadIdx.x];
    // 2. the wave's mean colour out of its pixel registers: every lane's own pixel (x + 1, y0 + 1), a 16 x 4 lattice over the block, summed
    // by DPP row additions (colorlut_xtile_kernel, anchor 7).  A block that sticks out of the frame offers its top-left pixel; one that lies
    // wholly outside offers nothing.  Bit 31 says "offered".
    uint32_t mine_mean = 0;
    if (bx + kTileW <= width && by + kTileH <= height) { // wave-uniform
        const uint32_t mine = v[MVFX_XTILE_SAMPLE_ROW < kRows ? MVFX_XTILE_SAMPLE_ROW : 0].y;
        uint32_t ev = mine & 0x00ff00ffu, od = (mine >> 8) & 0x00ff00ffu;
#define MVFX_ROW_ADD(v_, ctrl) v_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v_, ctrl, 0xf, 0xf, true)
        MVFX_ROW_ADD(ev, 0x111); MVFX_ROW_ADD(od, 0x111);
        MVFX_ROW_ADD(ev, 0x112); MVFX_ROW_ADD(od, 0x112);
        MVFX_ROW_ADD(ev, 0x114); MVFX_ROW_ADD(od, 0x114);
        MVFX_ROW_ADD(ev, 0x118); MVFX_ROW_ADD(od, 0x118);
#undef MVFX_ROW_ADD
        const uint32_t sev = (uint32_t)__builtin_amdgcn_readlane((int)ev, 15) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 31) +
                             (uint32_t)__builtin_amdgcn_readlane((int)ev, 47) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 63) + 0x00200020u;
        const uint32_t sod = (uint32_t)__builtin_amdgcn_readlane((int)od, 15) + (uint32_t)__builtin_amdgcn_readlane((int)od, 31) +
                             (uint32_t)__builtin_amdgcn_readlane((int)od, 47) + (uint32_t)__builtin_amdgcn_readlane((int)od, 63) + 0x00200020u;
        mine_mean = ((sev >> 6) & 0x00ff00ffu) | (((sod >> 6) & 0x000000ffu) << 8) | 0x80000000u;
    } else if (bx < width && by < height) {
        mine_mean = ((uint32_t)__builtin_amdgcn_readlane((int)v[0].x, 0) & 0xffffffu) | 0x80000000u;
    }
    if (lane == 0) wave_anchor[wave] = mine_mean;
    __syncthreads(); // the coordinate table and the four means
    // 3. the workgroup's window, anchored at the mean of the means on offer (1, 2 or 4 of them: waves drop out by column or by row)
    uint32_t ar, ayp, azp, ccpx;
    {
        uint32_t sev = 0, sod = 0, n = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_anchor[k]);
            if (a >> 31) {
                sev += a & 0x00ff00ffu;
                sod += (a >> 8) & 0x000000ffu;
                n++;
            }
        }
        const uint32_t sh = n == 4 ? 2u : n == 2 ? 1u : 0u, half = (1u << sh) >> 1; // (n == 3 cannot happen on a 2 x 2 grid; it would keep the sum of... guarded below)
        uint32_t cpx = (((sev + half * 0x00010001u) >> sh) & 0x00ff00ffu) | ((((sod + half) >> sh) & 0xffu) << 8);
        if (n == 3 || n == 0) cpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_anchor[0]) & 0xffffffu;
        ccpx = cpx;
        const uint32_t cr = cpx & 0xffu;
        // the anchor's lattice coordinates by arithmetic, not by two dependent scalar loads (colorlut_xtile_kernel)
        const float gy = (float)((cpx >> 8) & 0xffu) * (1.0f / 255.0f), bz = (float)((cpx >> 16) & 0xffu) * (1.0f / 255.0f);
        const float ny = fminf(fmaxf(gy * p.scale[1] + p.offset[1], 0.0f), 1.0f) * p.size_m1, nz = fminf(fmaxf(bz * p.scale[2] + p.offset[2], 0.0f), 1.0f) * p.size_m1;
        ar = min((cr > RW / 2 ? cr - RW / 2 : 0u) & ~1u, 256u - RW); // even: a window row starts on a 16-byte piece
        const uint32_t ay = (uint32_t)__builtin_amdgcn_readfirstlane((int)xtile_first_cell<kWgNY>(ny, p.size)),
                       az = (uint32_t)__builtin_amdgcn_readfirstlane((int)xtile_first_cell<kWgNZ>(nz, p.size)); // z rows run 0 .. size
        ayp = ay * kWgPitchY;
        azp = az * kWgPitchZ;
        // NY x NZR rows of RW entries of the x table, global -> LDS directly, 16-byte pieces (xtile_fill_window; here all four waves fill)
        typedef __attribute__((address_space(3))) void *lds_void_t;
        typedef const __attribute__((address_space(1))) void *global_void_t;
        constexpr uint32_t kRowP = RW * 3 / 2, kPieces = kWgNY * kWgNZR * kRowP;
        const uint32_t base = (ay * (p.size + 1) + az) * kXRowPieces + ar * 3 / 2; // workgroup-uniform
#pragma unroll
        for (uint32_t q0 = 0; q0 < kPieces; q0 += kBlock) {
            const uint32_t q = q0 + threadIdx.x;
            if (q0 + kBlock <= kPieces || q < kPieces) {
                const uint32_t wr = q / kRowP, k = q - wr * kRowP; // window row = dy * NZR + dz
                __builtin_amdgcn_global_load_lds((global_void_t)(p.xtable + (base + ((wr / kWgNZR) * (p.size + 1) + (wr % kWgNZR)) * kXRowPieces + k)),
                                                 (lds_void_t)(win + (q0 + wave * 64u) * 16u), 16, 0, 0);
            }
        }
    }
    __syncthreads(); // the window
    const uint32_t lds_k = 0u - ayp - azp - ar * 24u, ar24 = ar * 24u;
    // 4. the rows (colorlut_xtile_kernel's row pass; the outside pixel is patched from the x table in global memory in ONE branch per row)
#ifndef MVFX_XWG_PP
#define MVFX_XWG_PP 4 // pixels per pass of a row (2: two passes of two pixels, fewer live registers)
#endif
    constexpr int PP = MVFX_XWG_PP;
    auto do_row = [&](const uint32_t row) -> bool { // true: row 0 found the block "far" (uniform-random colours) -- nothing served, nothing stored
        const bool valid = x < width && y0 + row < height; // width % 4 == 0 (launcher)
        uint32_t px[4] = {v[row].x, v[row].y, v[row].z, v[row].w};
        f32x2_t e0[4][3], e1[4][3];
        float ty[4], tz[4];
        bool miss[4];
        bool any_miss = false;
        uint32_t outside = 0;
        if (PP == 2 && row == 0 && MVFX_XTILE_FAR_GATHER) { // the far test needs the whole row's misses: a cheap pass of its own (row 0 only)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t pxj = px[j];
                const uint2 eg = coord[(pxj >> 8) & 0xffu], eb = coord[256 + ((pxj >> 16) & 0xffu)];
                const uint32_t dr24 = (pxj & 0xffu) * 24u - ar24, dyp = eg.x - ayp, dzp = eb.x - azp;
                miss[j] = (dr24 >= RW * 24u) | (dyp >= kWgNY * kWgPitchY) | (dzp >= kWgNZ * kWgPitchZ);
                outside += (uint32_t)__popcll(__ballot(miss[j] & valid));
            }
        }
#pragma unroll
        for (int h = 0; h < 4 / PP; h++) {
        if (PP == 2 && h == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = h * PP; j < (h + 1) * PP; j++) {
            const uint32_t pxj = px[j];
            const uint2 eg = coord[(pxj >> 8) & 0xffu], eb = coord[256 + ((pxj >> 16) & 0xffu)];
            ty[j] = __uint_as_float(eg.y);
            tz[j] = __uint_as_float(eb.y);
            uint32_t r24;
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r24) : "v"(pxj), "v"(24u));
            const uint32_t dr24 = r24 - ar24, dyp = eg.x - ayp, dzp = eb.x - azp; // unsigned: below the anchor wraps to a huge value
            miss[j] = (dr24 >= RW * 24u) | (dyp >= kWgNY * kWgPitchY) | (dzp >= kWgNZ * kWgPitchZ);
            any_miss = any_miss | miss[j];
            if (PP == 4 && MVFX_XTILE_FAR_GATHER && row == 0) outside += (uint32_t)__popcll(__ballot(miss[j] & valid));
            const uint32_t off = miss[j] ? 0u : eg.x + eb.x + (r24 + lds_k);
            const lds_float2_t q0 = (lds_float2_t)((lds_bytes_t)&win[0] + off), q1 = (lds_float2_t)((lds_bytes_t)&win[0] + off + kWgPitchZ);
            e0[j][0] = q0[0]; e0[j][1] = q0[1]; e0[j][2] = q0[2];
            e1[j][0] = q1[0]; e1[j][1] = q1[1]; e1[j][2] = q1[2];
        }
        if (MVFX_XTILE_FAR_GATHER && row == 0 && h == 0 && outside > 248u) { // wave-uniform, rare (colorlut_xtile_kernel has the reasoning)
            uint32_t far = 0, alike = 0, fpx = 0;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t b = __ballot(miss[j] & valid);
                if (b != 0 && !found) {
                    fpx = (uint32_t)__builtin_amdgcn_readlane((int)px[j], __builtin_ctzll(b)); // the first outside pixel
                    found = true;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                far += (uint32_t)__popcll(__ballot(miss[j] & valid & (__builtin_amdgcn_sad_u8(px[j] & 0x00ffff00u, ccpx & 0x00ffff00u, 0u) > (uint32_t)MVFX_XTILE_FAR)));
                alike += (uint32_t)__popcll(__ballot(miss[j] & valid & (((px[j] ^ fpx) & 0x00f0f000u) == 0u)));
            }
            if (far * 4u > outside * 3u && alike * 4u < outside) return true;
        }
        if (any_miss) {
#pragma unroll
            for (int j = h * PP; j < (h + 1) * PP; j++) {
                if (miss[j]) {
                    const uint32_t pxj = px[j];
                    const uint32_t iy = coord[(pxj >> 8) & 0xffu].x / kWgPitchY, iz = coord[256 + ((pxj >> 16) & 0xffu)].x / kWgPitchZ, r = pxj & 0xffu;
                    const f32x2_t *g0p = reinterpret_cast<const f32x2_t *>(p.xtable) + (uint64_t)((iy * (p.size + 1) + iz) * 256u + r) * 3, *g1p = g0p + 256 * 3;
                    e0[j][0] = g0p[0]; e0[j][1] = g0p[1]; e0[j][2] = g0p[2];
                    e1[j][0] = g1p[0]; e1[j][1] = g1p[1]; e1[j][2] = g1p[2];
                }
            }
        }
#pragma unroll
        for (int j = h * PP; j < (h + 1) * PP; j++) {
            // entry = (X.r, X.g) (X.b, D.r) (D.g, D.b)
            const float c0r = e0[j][0].x + e0[j][1].y * ty[j], c0g = e0[j][0].y + e0[j][2].x * ty[j], c0b = e0[j][1].x + e0[j][2].y * ty[j];
            const float c1r = e1[j][0].x + e1[j][1].y * ty[j], c1g = e1[j][0].y + e1[j][2].x * ty[j], c1b = e1[j][1].x + e1[j][2].y * ty[j];
            const float rr = lf_add_clamp(c0r, (c1r - c0r) * tz[j]), gg = lf_add_clamp(c0g, (c1g - c0g) * tz[j]),
                        bb = lf_add_clamp(c0b, (c1b - c0b) * tz[j]);
            // float_to_u8 (imp.rs:537-539) as ONE fused multiply-add + truncation (tools/prove_exact.c P15, exhaustive)
            const float yr = __builtin_fmaf(rr, p.fast.out_scale, p.fast.pred_half), yg = __builtin_fmaf(gg, p.fast.out_scale, p.fast.pred_half),
                        yb = __builtin_fmaf(bb, p.fast.out_scale, p.fast.pred_half);
            uint32_t w = px[j];
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yr));
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yg));
            asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(yb));
            px[j] = w;
        }
        if (PP == 2) any_miss = false; // per pass
        } // passes
        if (valid) {
            u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)row * out_stride + voff_out);
            const u32x4_t t = {px[0], px[1], px[2], px[3]};
            if (MVFX_XTILE_NT) __builtin_nontemporal_store(t, dst);
            else *dst = t;
        }
        return false;
    };
    if (do_row(0)) { // wave-uniform, rare: every lane gathers its own pixels' cells (no barrier follows: the other waves go on)
        CellCache cache;
#pragma unroll
        for (uint32_t hr = 0; hr < kRows; hr++) {
            uint4 q = v[hr];
            q.x = lf_px8<true, true>(q.x, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.y = lf_px8<true, true>(q.y, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.z = lf_px8<true, true>(q.z, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            q.w = lf_px8<true, true>(q.w, p, p.cube, p.t[0], p.t[1], p.t[2], cache);
            if (x < width && y0 + hr < height) {
                u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)hr * out_stride + voff_out);
                const u32x4_t t = {q.x, q.y, q.z, q.w};
                if (MVFX_XTILE_NT) __builtin_nontemporal_store(t, dst);
                else *dst = t;
            }
        }
        return;
    }
#pragma unroll
    for (uint32_t row = 1; row < kRows; row++) do_row(row);
}

// ---------------------------------------------------------------- content probe: which window kernel suits the stream (round 5)
//
// colorlut_xtile_kernel (an 18 x 3 x 3 window per wave, six workgroups per CU) is the faster kernel on calm pictures -- 16 x 4K per launch,
// same box: 80-82 k fps on smooth gradients, 78-79 k with +-3 codes of noise, 69-70 k with +-5 -- and collapses where the colours of a
// 64 x 20 block scatter: 41 k at +-8.  colorlut_xwg_kernel (one 38 x 5 x 5 window per workgroup) runs 74 / 72 / 70.5 / 69 / 47 k at
// +-0 / 3 / 5 / 8 / 16 on the same frames (profiles/r5/colorlut_experiments.txt).  The pictures of a stream resemble their predecessors, so the choice is made from a look at
// an earlier frame: one workgroup, 256 blocks of 64 x 20 pixels spread over the frame, sixteen pixels of each (a 4 x 4 lattice); a block
// is BUSY when the sampled bytes of a channel span more than kProbeSpan codes (sixteen samples of +-5 codes of noise on a gradient span
// about 15, of +-8 about 20).  More than kProbeBusy busy blocks of 256 make the picture busy.  Every thread of the launch writes nothing
// but thread 0, which stores the verdict into page-locked host memory; the launcher reads that word whenever it launches -- never
// waiting for it -- and runs the probe again every kProbeEvery launches.  Both kernels produce the same bytes: the verdict only moves time.
constexpr uint32_t kProbeSpan = 17, kProbeBusy = 38, kProbeEvery = 32;

__global__ __launch_bounds__(256) void colorlut_probe_kernel(const uint8_t *__restrict__ frame, uint32_t width, uint32_t height, uint32_t stride,
                                                            uint32_t *__restrict__ verdict)
{
    __shared__ uint32_t busy_blocks;
    if (threadIdx.x == 0) busy_blocks = 0;
    __syncthreads();
    const uint32_t tiles_x = width / 64u, tiles_y = height / 20u; // whole blocks only; (0, 0) when the frame is smaller than one
    bool busy = false;
    if (tiles_x != 0 && tiles_y != 0) {
        // block (i, j) of a 16 x 16 lattice over the whole blocks of the frame
        const uint32_t tx = (uint32_t)(((uint64_t)(threadIdx.x & 15u) * 2u + 1u) * tiles_x / 32u), ty = (uint32_t)(((uint64_t)(threadIdx.x >> 4) * 2u + 1u) * tiles_y / 32u);
        uint32_t lo[3] = {255u, 255u, 255u}, hi[3] = {0u, 0u, 0u};
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t px = *reinterpret_cast<const uint32_t *>(frame + (size_t)(ty * 20u + 2u + 5u * (k >> 2)) * stride + (tx * 64u + 8u + 16u * (k & 3u)) * 4u);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t b = (px >> (8 * c)) & 0xffu;
                lo[c] = min(lo[c], b);
                hi[c] = max(hi[c], b);
            }
        }
        busy = max(max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) > kProbeSpan;
    }
    const uint32_t n = (uint32_t)__popcll(__ballot(busy));
    if ((threadIdx.x & 63u) == 0 && n != 0) atomicAdd(&busy_blocks, n);
    __syncthreads();
    if (threadIdx.x == 0) {
        verdict[1] = busy_blocks;
        __atomic_store_n(&verdict[0], busy_blocks > kProbeBusy ? 2u : 1u, __ATOMIC_RELAXED);
    }
}

// ---------------------------------------------------------------- colorlut on I420 frames, fused
//
// `videoconvert ! colorlut ! videoconvert` of the reference's example pipeline (colorlut/imp.rs:17-19) in ONE kernel:
// I420 -> RGBA (convert_math.hpp), the FAST LUT path above, RGBA -> I420, with the two RGBA frames never leaving the
// registers: 1.5 B/px read + 1.5 B/px written instead of 1.5+4 | 4+4 | 4+1.5 = 19 B/px through three launches.
// The tile walk (i420_fused_tile) is shared with hsvfilter's I420 entry point: convert_math.hpp.
template <bool IS3D, bool CELLS>
__global__ __launch_bounds__(kI420Block) void colorlut_i420_kernel(I420Planes pl, uint32_t width, uint32_t height, LutParams p,
                                                                   YuvToRgbCoef kin, RgbToYuvCoef kout)
{
    __shared__ int2 edge[kI420Block];
    CellCache cache;
    i420_fused_tile(pl, width, height, kin, kout, edge,
                    [&](uint32_t px) { return lf_px8<IS3D, CELLS>(px, p, p.cube, p.t[0], p.t[1], p.t[2], cache); });
}

// The fused I420 kernel with the wave-local window: the compact walk of convert_math.hpp (a wave = 64 x 16 pixels, each lane an
// 8 x 2 strip of it), the window anchored at the cell of the wave's centre pixel (the first pixel of lane 36 = column 32, row 8 of the
// block; lane 0's when the centre lies outside the frame), coordinates from the byte table.  On natural-like content the per-lane
// gathers of colorlut_i420_kernel were the bound: 38.7 us per 4K frame against 31.7 us on one flat colour (no gathers at all;
// 27.7 us with the coordinates from the byte table).  This kernel: 30.5 us natural-like, 29.4 us flat.
__global__ __launch_bounds__(kI420Block) void colorlut_i420_tile_kernel(I420Planes pl, uint32_t width, uint32_t height, LutParams p,
                                                                        YuvToRgbCoef kin, RgbToYuvCoef kout)
{
    __shared__ int2 edge[kI420Block];
    __shared__ uint2 coord[kCoordEntries];
    __shared__ float4 nbr[kI420Block / 64][kTileWaveLdsFloat4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        const uint2 *src = reinterpret_cast<const uint2 *>(p.tile_tables);
#pragma unroll
        for (uint32_t i = 0; i < kCoordEntries / kI420Block; i++) coord[i * kI420Block + threadIdx.x] = src[i * kI420Block + threadIdx.x];
    }
    uint32_t x0, y0, edge_index;
    bool has_left;
    i420_lane_origin<true>(x0, y0, edge_index, has_left);
    const bool active = x0 < width && y0 < height;
    uint32_t first = 0xff000000u;
    if (active) {
        const uint32_t crow = y0 / 2;
        const ChromaTerms c = chroma_terms(pl.iu[(uint64_t)crow * pl.ius + x0 / 2], pl.iv[(uint64_t)crow * pl.ivs + x0 / 2], kin);
        first = yuv_pixel(pl.iy[(uint64_t)y0 * pl.iys + x0], c, kin);
    }
    const TileRel rel = tile_rel(lane, p);
    __syncthreads(); // coordinate table complete
    const uint32_t centre = __builtin_amdgcn_readlane((int)active, 36) ? 36u : 0u;
    const uint32_t fpx = (uint32_t)__builtin_amdgcn_readlane((int)first, centre);
    const uint32_t cx = coord[fpx & 0xffu].x, cy = coord[256 + ((fpx >> 8) & 0xffu)].x, cz = coord[512 + ((fpx >> 16) & 0xffu)].x;
    uint32_t ax, ay, az;
    tile_load_window(nbr[wave], lane, p, rel, cx, cy, cz, ax, ay, az);
    const uint32_t wave_lds_bytes = wave * (uint32_t)(kTileWaveLdsFloat4 * sizeof(float4));
    i420_fused_tile<true>(pl, width, height, kin, kout, edge, [&](uint32_t px) {
        const uint2 er = coord[px & 0xffu], eg = coord[256 + ((px >> 8) & 0xffu)], eb = coord[512 + ((px >> 16) & 0xffu)];
        float4 c[8];
        tile_cell((lds_bytes_t)&nbr[0][0], wave_lds_bytes, p, er.x, eg.x, eb.x, ax, ay, az, c);
        float r, g, b;
        lf_trilinear<true>(c, __uint_as_float(er.y), __uint_as_float(eg.y), __uint_as_float(eb.y), r, g, b);
        const float yr = r * p.fast.out_scale + p.fast.pred_half, yg = g * p.fast.out_scale + p.fast.pred_half,
                    yb = b * p.fast.out_scale + p.fast.pred_half;
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yr));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yg));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yb));
        return px;
    });
}

// The fused I420 kernel on the x-prelerped table (round 3): the compact walk of convert_math.hpp as above, the LUT step of
// colorlut_xtile_kernel -- window of MVFX_XTILE_RW r bytes x 3 y cells x 4 z rows around the first pixel of lane 36 (lane 0's when
// the block's centre lies outside the frame), filled by global_load_lds, premultiplied coordinate entries for g and b.
__global__ __launch_bounds__(kI420Block) void colorlut_i420_xtile_kernel(I420Planes pl, uint32_t width, uint32_t height, LutParams p,
                                                                         YuvToRgbCoef kin, RgbToYuvCoef kout)
{
    constexpr uint32_t RW = MVFX_XTILE_RW, kWaveBytes = kXWaveBytes;
    __shared__ int2 edge[kI420Block];
    __shared__ uint2 coord[512];
    __shared__ __attribute__((aligned(16))) uint8_t win[(kI420Block / 64) * kWaveBytes];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 512; i += kI420Block) coord[i] = p.xcoord[i];
    uint32_t x0, y0, edge_index;
    bool has_left;
    i420_lane_origin<true>(x0, y0, edge_index, has_left);
    const bool active = x0 < width && y0 < height;
    uint32_t first = 0xff000000u;
    if (active) {
        const uint32_t crow = y0 / 2;
        const ChromaTerms c = chroma_terms(pl.iu[(uint64_t)crow * pl.ius + x0 / 2], pl.iv[(uint64_t)crow * pl.ivs + x0 / 2], kin);
        first = yuv_pixel(pl.iy[(uint64_t)y0 * pl.iys + x0], c, kin);
    }
    const uint32_t centre = __builtin_amdgcn_readlane((int)active, 36) ? 36u : 0u;
    uint32_t fpx = (uint32_t)__builtin_amdgcn_readlane((int)first, centre);
#if MVFX_I420_XTILE_MEAN
    // a block that lies wholly inside the frame is anchored at the MEAN of its 64 lanes' first pixels (an 8 x 8 lattice over the
    // 64 x 16 block, already in registers) unless its corners say an edge runs through it: colorlut_xtile_kernel, MVFX_XTILE_ANCHOR4 7
    if (__ballot(active) == ~0ull) {
        uint32_t ev = first & 0x00ff00ffu, od = (first >> 8) & 0x00ff00ffu;
#define MVFX_ROW_ADD(v_, ctrl) v_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v_, ctrl, 0xf, 0xf, true)
        MVFX_ROW_ADD(ev, 0x111); MVFX_ROW_ADD(od, 0x111);
        MVFX_ROW_ADD(ev, 0x112); MVFX_ROW_ADD(od, 0x112);
        MVFX_ROW_ADD(ev, 0x114); MVFX_ROW_ADD(od, 0x114);
        MVFX_ROW_ADD(ev, 0x118); MVFX_ROW_ADD(od, 0x118);
#undef MVFX_ROW_ADD
        const uint32_t sev = (uint32_t)__builtin_amdgcn_readlane((int)ev, 15) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 31) +
                             (uint32_t)__builtin_amdgcn_readlane((int)ev, 47) + (uint32_t)__builtin_amdgcn_readlane((int)ev, 63) + 0x00200020u;
        const uint32_t sod = (uint32_t)__builtin_amdgcn_readlane((int)od, 15) + (uint32_t)__builtin_amdgcn_readlane((int)od, 31) +
                             (uint32_t)__builtin_amdgcn_readlane((int)od, 47) + (uint32_t)__builtin_amdgcn_readlane((int)od, 63) + 0x00200020u;
        const uint32_t mean = ((sev >> 6) & 0x00ff00ffu) | (((sod >> 6) & 0x000000ffu) << 8);
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)first, 0) & 0xffffffu, q1 = (uint32_t)__builtin_amdgcn_readlane((int)first, 7) & 0xffffffu,
                       q2 = (uint32_t)__builtin_amdgcn_readlane((int)first, 56) & 0xffffffu, q3 = (uint32_t)__builtin_amdgcn_readlane((int)first, 63) & 0xffffffu;
        if (__builtin_amdgcn_sad_u8(q0, q3, 0u) + __builtin_amdgcn_sad_u8(q1, q2, 0u) <= (uint32_t)MVFX_XTILE_SPREAD64) fpx = mean;
    }
#endif
    const uint32_t cr = fpx & 0xffu, cy = p.tile_tables[2 * (256 + ((fpx >> 8) & 0xffu))], cz = p.tile_tables[2 * (512 + ((fpx >> 16) & 0xffu))];
    const uint32_t ar = min((cr > RW / 2 ? cr - RW / 2 : 0u) & ~1u, 256u - RW);
    const uint32_t ay = min(cy > (kXNY - 1) / 2 ? cy - (kXNY - 1) / 2 : 0u, p.size - kXNY), az = min(cz > (kXNZ - 1) / 2 ? cz - (kXNZ - 1) / 2 : 0u, p.size - kXNZ);
    const uint32_t ayp = ay * kXPitchY, azp = az * kXPitchZ, ar24 = ar * 24u;
    {
        const uint32_t base = (ay * (p.size + 1) + az) * kXRowPieces + ar * 3 / 2;
        xtile_fill_window(p.xtable, base, p.size, win + wave * kWaveBytes, lane);
    }
    __syncthreads(); // coordinate table and window complete
    const uint32_t wave_lds = wave * kWaveBytes, lds_k = wave_lds - ayp - azp - ar24;
    i420_fused_tile<true>(pl, width, height, kin, kout, edge, [&](uint32_t px) {
        const uint2 eg = coord[(px >> 8) & 0xffu], eb = coord[256 + ((px >> 16) & 0xffu)];
        const float ty = __uint_as_float(eg.y), tz = __uint_as_float(eb.y);
        uint32_t r24;
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r24) : "v"(px), "v"(24u));
        const uint32_t dr24 = r24 - ar24, dyp = eg.x - ayp, dzp = eb.x - azp;
        const bool miss = (dr24 >= RW * 24u) | (dyp >= kXNY * kXPitchY) | (dzp >= kXNZ * kXPitchZ);
        const uint32_t off = miss ? wave_lds : eg.x + eb.x + (r24 + lds_k);
        const lds_float2_t q0 = (lds_float2_t)((lds_bytes_t)&win[0] + off), q1 = (lds_float2_t)((lds_bytes_t)&win[0] + off + kXPitchZ);
        f32x2_t e0[3] = {q0[0], q0[1], q0[2]}, e1[3] = {q1[0], q1[1], q1[2]};
        if (miss) {
            const uint32_t iy = eg.x / kXPitchY, iz = eb.x / kXPitchZ, r = px & 0xffu;
            const f32x2_t *g0 = reinterpret_cast<const f32x2_t *>(p.xtable) + (uint64_t)((iy * (p.size + 1) + iz) * 256u + r) * 3, *g1 = g0 + 256 * 3;
            e0[0] = g0[0]; e0[1] = g0[1]; e0[2] = g0[2];
            e1[0] = g1[0]; e1[1] = g1[1]; e1[2] = g1[2];
        }
        const float c0r = e0[0].x + e0[1].y * ty, c0g = e0[0].y + e0[2].x * ty, c0b = e0[1].x + e0[2].y * ty;
        const float c1r = e1[0].x + e1[1].y * ty, c1g = e1[0].y + e1[2].x * ty, c1b = e1[1].x + e1[2].y * ty;
        const float rr = lf_add_clamp(c0r, (c1r - c0r) * tz), gg = lf_add_clamp(c0g, (c1g - c0g) * tz), bb = lf_add_clamp(c0b, (c1b - c0b) * tz);
        const float yr = __builtin_fmaf(rr, p.fast.out_scale, p.fast.pred_half), yg = __builtin_fmaf(gg, p.fast.out_scale, p.fast.pred_half),
                    yb = __builtin_fmaf(bb, p.fast.out_scale, p.fast.pred_half); // P15
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yr));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yg));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(px) : "v"(yb));
        return px;
    });
}

template <bool IS3D, bool CELLS, bool WIDE, bool LE>
__global__ __launch_bounds__(kBlock) void colorlut_fast_global_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width,
                                                                      uint32_t rows, uint64_t in_stride,
                                                                      uint64_t out_stride, LutParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z]; // one frame pair of the batch per grid z
    uint8_t *out = out_fb.base[blockIdx.z];
    lf_rows<IS3D, CELLS, WIDE, LE>(in, out, width, rows, in_stride, out_stride, p, p.cube, p.t[0], p.t[1], p.t[2],
                                   blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, blockIdx.y, gridDim.y);
}

template <bool IS3D, bool WIDE, bool LE>
__global__ __launch_bounds__(kLdsBlock) void colorlut_fast_lds_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width,
                                                                      uint32_t rows, uint64_t in_stride,
                                                                      uint64_t out_stride, LutParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z]; // one frame pair of the batch per grid z
    uint8_t *out = out_fb.base[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    if constexpr (IS3D) {
        float4 *cube = reinterpret_cast<float4 *>(lds_raw);
        const uint32_t n = p.size * p.size * p.size;
        for (uint32_t i = threadIdx.x; i < n; i += kLdsBlock)
            cube[i] = p.cube[i];
        __syncthreads();
        lf_rows<IS3D, false, WIDE, LE>(in, out, width, rows, in_stride, out_stride, p, (const float4 *)cube,
                                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                                       blockIdx.x * kLdsBlock + threadIdx.x, gridDim.x * kLdsBlock, blockIdx.y, gridDim.y);
    } else {
        float *t = reinterpret_cast<float *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < p.size; i += kLdsBlock) {
            t[i] = p.t[0][i];
            t[p.size + i] = p.t[1][i];
            t[2 * p.size + i] = p.t[2][i];
        }
        __syncthreads();
        lf_rows<IS3D, false, WIDE, LE>(in, out, width, rows, in_stride, out_stride, p, (const float4 *)nullptr,
                                       (const float *)t, (const float *)(t + p.size), (const float *)(t + 2 * p.size),
                                       blockIdx.x * kLdsBlock + threadIdx.x, gridDim.x * kLdsBlock, blockIdx.y, gridDim.y);
    }
}


int ensure_uploaded(mvfx_cube_lut *h)
{
    int dev = 0;
    MVFX_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(h->mu);
    if (h->device == dev && (h->d_rgba || h->d_table[0]))
        return MVFX_OK;
    // (re)upload for this device
    if (h->d_rgba) { (void)hipFree(h->d_rgba); h->d_rgba = nullptr; }
    if (h->d_cells) { (void)hipFree(h->d_cells); h->d_cells = nullptr; }
    if (h->d_xtable) { (void)hipFree(h->d_xtable); h->d_xtable = nullptr; }
    if (h->d_xcoord) { (void)hipFree(h->d_xcoord); h->d_xcoord = nullptr; }
    if (h->d_xcoord_wg) { (void)hipFree(h->d_xcoord_wg); h->d_xcoord_wg = nullptr; }
    if (h->d_tile_tables) { (void)hipFree(h->d_tile_tables); h->d_tile_tables = nullptr; }
    if (h->d_baked) { (void)hipFree(h->d_baked); h->d_baked = nullptr; }
    for (auto &t : h->d_table) if (t) { (void)hipFree(t); t = nullptr; }
    const CubeLut &l = h->lut;
    if (l.is_3d) {
        const size_t bytes = l.rgba.size() * sizeof(float);
        MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_rgba), bytes));
        MVFX_HIP_TRY(hipMemcpy(h->d_rgba, l.rgba.data(), bytes, hipMemcpyHostToDevice));
        if (l.size <= kCellMaxSize) { // cell-packed copy: corner (i,j,k) of cell (x,y,z) = node(min(x+i,m), ...)
            const size_t n = (size_t)l.size, m = n - 1;
            std::vector<float> cells(n * n * n * kCellF4 * 4, 0.0f);
            for (size_t z = 0; z < n; z++)
                for (size_t y = 0; y < n; y++)
                    for (size_t x = 0; x < n; x++)
                        for (size_t c = 0; c < 8; c++) {
                            const size_t xx = std::min(x + (c & 1), m), yy = std::min(y + ((c >> 1) & 1), m), zz = std::min(z + (c >> 2), m);
                            std::memcpy(&cells[(x + n * (y + n * z)) * (kCellF4 * 4) + c * 3], &l.rgba[(xx + n * (yy + n * zz)) * 4], 12);
                        }
            // odd corners (x+1) become the x-differences RN(c_odd - c_even): what `a + (b - a) * t` subtracts per pixel
            for (size_t cell = 0; cell < n * n * n; cell++)
                for (size_t pair = 0; pair < 4; pair++)
                    for (size_t ch = 0; ch < 3; ch++) {
                        float *even = &cells[cell * (kCellF4 * 4) + 2 * pair * 3 + ch], *odd = even + 3;
                        *odd = *odd - *even;
                    }
            MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_cells), cells.size() * sizeof(float)));
            MVFX_HIP_TRY(hipMemcpy(h->d_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
            // tables of colorlut_tile_kernel: per channel and byte value the lattice cell index and fraction -- norm_comp *
            // (size - 1), floor, t = x - x0 (imp.rs:471-474, 496-506) evaluated here in the same f32 steps the FAST device
            // functions take (this file is built with -ffp-contract=off; the domain is finite on that path) -- and the
            // offsets of the 162 sixteen-byte pieces of a 3 x 3 x 3 cell neighbourhood relative to its anchor cell
            if (l.size >= 3) {
                std::vector<uint32_t> tt(3 * 256 * 2 + 192, 0u);
                const float size_m1 = (float)l.size - 1.0f;
                for (int c = 0; c < 3; c++)
                    for (int b = 0; b < 256; b++) {
                        const float v = (float)b / 255.0f;                       // RN(b / 255) == the device's mul + fmac form (P8)
                        float n = v * l.domain_scale[c];
                        n = n + l.domain_offset[c];
                        n = n < 0.0f ? 0.0f : (n > 1.0f ? 1.0f : n);             // v_add_f32 clamp on finite values
                        const float x = n * size_m1;
                        const uint32_t i0 = (uint32_t)x;                         // floor (x >= 0)
                        const float t = x - (float)i0;
                        tt[(c * 256 + b) * 2] = i0;
                        std::memcpy(&tt[(c * 256 + b) * 2 + 1], &t, 4);
                    }
                for (uint32_t q = 0; q < 162; q++) {
                    const uint32_t run = q / 18u, piece = q - run * 18u, dz = run / 3u, dy = run - dz * 3u;
                    tt[1536 + q] = (l.size * (dy + l.size * dz) + piece / 6u) * kCellF4 + piece % 6u; // float4 units; run = dy + 3 dz: three x-adjacent cells
                }
                MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_tile_tables), tt.size() * sizeof(uint32_t)));
                MVFX_HIP_TRY(hipMemcpy(h->d_tile_tables, tt.data(), tt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                if (l.size >= 4) { // the x-prelerped table of colorlut_xtile_kernel, computed on the device from the two copies above
                    const size_t entries = (size_t)l.size * (l.size + 1) * 256;
                    MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_xtable), entries * 6 * sizeof(float)));
                    MVFX_LAUNCH(colorlut_xtable_build_kernel, dim3((uint32_t)((entries + 255) / 256)), dim3(256), 0, nullptr,
                                       reinterpret_cast<const float4 *>(h->d_rgba), h->d_tile_tables, l.size, h->d_xtable);
                    MVFX_HIP_TRY(hipGetLastError());
                    std::vector<uint32_t> xc(512 * 2);
                    for (int b = 0; b < 256; b++) {
                        xc[2 * b] = tt[(256 + b) * 2] * kXPitchY;          xc[2 * b + 1] = tt[(256 + b) * 2 + 1];
                        xc[2 * (256 + b)] = tt[(512 + b) * 2] * kXPitchZ;  xc[2 * (256 + b) + 1] = tt[(512 + b) * 2 + 1];
                    }
                    MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_xcoord), xc.size() * sizeof(uint32_t)));
                    MVFX_HIP_TRY(hipMemcpy(h->d_xcoord, xc.data(), xc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    if (l.size >= kWgNY && l.size >= kWgNZ) { // colorlut_xwg_kernel's window is 5 x 5 cells
                        for (int b = 0; b < 256; b++) {
                            xc[2 * b] = tt[(256 + b) * 2] * kWgPitchY;
                            xc[2 * (256 + b)] = tt[(512 + b) * 2] * kWgPitchZ;
                        }
                        MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_xcoord_wg), xc.size() * sizeof(uint32_t)));
                        MVFX_HIP_TRY(hipMemcpy(h->d_xcoord_wg, xc.data(), xc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                        if (!h->h_probe) { // (without it the automatic choice is always the workgroup-window kernel)
                            void *q = nullptr;
                            if (hipHostMalloc(&q, 64, hipHostMallocDefault) == hipSuccess) {
                                std::memset(q, 0, 64);
                                h->h_probe = static_cast<uint32_t *>(q);
                            }
                        }
                    }
                    MVFX_HIP_TRY(hipStreamSynchronize(nullptr));
                }
            }
        }
    } else {
        for (int c = 0; c < 3; c++) {
            const size_t bytes = l.table[c].size() * sizeof(float);
            MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_table[c]), bytes));
            MVFX_HIP_TRY(hipMemcpy(h->d_table[c], l.table[c].data(), bytes, hipMemcpyHostToDevice));
        }
    }
    h->device = dev;
    return MVFX_OK;
}

template <bool IS3D, bool WIDE, bool LE, bool VEC>
int launch_one(bool use_lds, dim3 grid, size_t lds_bytes, hipStream_t st, const FrameBatch &in, const FrameBatch &out,
               uint64_t width, uint32_t rows, uint64_t is, uint64_t os, const LutParams &p)
{
    if (use_lds) {
        auto k = colorlut_lds_kernel<IS3D, WIDE, LE, VEC>;
        MVFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
        MVFX_LAUNCH(k, grid, dim3(kLdsBlock), lds_bytes, st, in, out, width, rows, is, os, p);
    } else {
        MVFX_LAUNCH((colorlut_global_kernel<IS3D, WIDE, LE, VEC>), grid, dim3(kBlock), 0, st, in, out, width,
                           rows, is, os, p);
    }
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

// ---- baked table (placement 6) --------------------------------------------------------------------------------------------
// An RGBA8 pixel is a pure function of its three colour bytes, so the whole LUT fits a table of 2^24 dwords: 64 MiB, a quarter of the
// 256 MiB Infinity Cache, nothing beside 288 GB of HBM.  Per pixel: one 4-byte gather instead of six 16-byte LDS reads and 57 f32
// operations; the interpolating kernels are VALU-bound, this one is bound by the L1 tag rate and by how many table lines
// a frame's colours touch.
constexpr uint32_t kBakedSide = 4096; // the all-colours frame: 4096 x 4096 pixels, pixel i holds colour i

__global__ __launch_bounds__(256) void colorlut_all_colours_kernel(uint32_t *frame)
{
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u;
    *reinterpret_cast<uint4 *>(frame + i) = make_uint4(i, i + 1, i + 2, i + 3); // alpha byte 0
}

// PER_LANE 16-byte groups per lane, all loads issued before the first gather (memory-level parallelism for the table misses)
template <int PER_LANE>
__global__ __launch_bounds__(256) void colorlut_baked_kernel(FrameBatch in, FrameBatch out, uint64_t vecs_per_row, uint32_t rows, uint64_t is,
                                                             uint64_t os, const uint32_t *__restrict__ table)
{
    const uint8_t *src = in.base[blockIdx.z];
    uint8_t *dst = out.base[blockIdx.z];
    const uint64_t x0 = ((uint64_t)blockIdx.x * PER_LANE) * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < rows; y += gridDim.y) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t *srow = reinterpret_cast<const u32x4_t *>(src + (uint64_t)y * is);
        u32x4_t *drow = reinterpret_cast<u32x4_t *>(dst + (uint64_t)y * os);
        u32x4_t v[PER_LANE];
#pragma unroll
        for (int k = 0; k < PER_LANE; k++) {
            const uint64_t x = x0 + (uint64_t)k * 256u;
            if (x < vecs_per_row) v[k] = __builtin_nontemporal_load(srow + x);
        }
#pragma unroll
        for (int k = 0; k < PER_LANE; k++) {
            const uint64_t x = x0 + (uint64_t)k * 256u;
            if (x < vecs_per_row) {
                u32x4_t o;
                o.x = table[v[k].x & 0xffffffu] | (v[k].x & 0xff000000u); // alpha byte copied (imp.rs:262, :291)
                o.y = table[v[k].y & 0xffffffu] | (v[k].y & 0xff000000u);
                o.z = table[v[k].z & 0xffffffu] | (v[k].z & 0xff000000u);
                o.w = table[v[k].w & 0xffffffu] | (v[k].w & 0xff000000u);
                __builtin_nontemporal_store(o, drow + x);
            }
        }
    }
}

int colorlut_impl(mvfx_cube_lut *h, const mvfx_frame *ins, const mvfx_frame *outs, uint32_t n, hipStream_t st, bool baking);

// Builds the table on first use (per LUT and device): all 2^24 colours through the interpolating kernels, then the host waits once.
int ensure_baked(mvfx_cube_lut *h, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(h->bake_mu);
    {
        std::lock_guard<std::mutex> l2(h->mu);
        if (h->d_baked) return MVFX_OK;
    }
    const size_t bytes = (size_t)kBakedSide * kBakedSide * 4;
    uint32_t *all = nullptr, *table = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&all), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&table), bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (all) (void)hipFree(all);
        return fail(MVFX_ERR_OUT_OF_MEMORY, "colorlut: no memory for the baked table (2 x 64 MiB)");
    }
    MVFX_LAUNCH(colorlut_all_colours_kernel, dim3(kBakedSide * kBakedSide / 1024), dim3(256), 0, st, all);
    mvfx_frame fi{}, fo{};
    fi.data = all; fo.data = table;
    fi.width = fo.width = kBakedSide; fi.height = fo.height = kBakedSide;
    fi.stride = fo.stride = kBakedSide * 4;
    fi.format = fo.format = MVFX_FORMAT_RGBA;
    int rc = colorlut_impl(h, &fi, &fo, 1, st, true);
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(all);
    if (rc == MVFX_OK && e != hipSuccess) rc = fail(MVFX_ERR_DEVICE, "colorlut: building the baked table failed: %s", hipGetErrorString(e));
    if (rc != MVFX_OK) { (void)hipFree(table); return rc; }
    std::lock_guard<std::mutex> l2(h->mu);
    h->d_baked = table;
    return MVFX_OK;
}

// n frame pairs sharing geometry and format through one LUT (n == 1: the reference's transform_frame)
int colorlut_impl(mvfx_cube_lut *h, const mvfx_frame *ins, const mvfx_frame *outs, uint32_t n, hipStream_t st, bool baking = false)
{
    if (!h)
        return fail(MVFX_ERR_NO_LUT, "colorlut: No LUT configured (colorlut/imp.rs:209-213)");
    if (!ins || !outs || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: NULL frame or empty batch");
    if (n > (uint32_t)kMaxBatch) { // split into launches of <= kMaxBatch pairs
        for (uint32_t done = 0; done < n; done += kMaxBatch) {
            const uint32_t m = (n - done) < (uint32_t)kMaxBatch ? (n - done) : (uint32_t)kMaxBatch;
            if (int rc = colorlut_impl(h, ins + done, outs + done, m, st, baking); rc != MVFX_OK) return rc;
        }
        return MVFX_OK;
    }
    const mvfx_frame *in = &ins[0], *out = &outs[0];
    for (uint32_t i = 1; i < n; i++) {
        if (ins[i].width != in->width || ins[i].height != in->height || ins[i].stride != in->stride || ins[i].format != in->format ||
            outs[i].width != out->width || outs[i].height != out->height || outs[i].stride != out->stride || outs[i].format != out->format)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: frames of one batch must share geometry and format");
        if (int rc = check_packed_frame(&ins[i], "colorlut input"); rc != MVFX_OK) return rc;
        if (int rc = check_packed_frame(&outs[i], "colorlut output"); rc != MVFX_OK) return rc;
    }
    if (in->format != MVFX_FORMAT_RGBA && in->format != MVFX_FORMAT_RGBA64_LE && in->format != MVFX_FORMAT_RGBA64_BE &&
        in->format != MVFX_FORMAT_RGB10A2_LE)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "colorlut: format %d is not RGBA / RGBA64_LE / RGBA64_BE (colorlut/imp.rs:122-134) or RGB10A2_LE "
                    "(d3d12colorlut/imp.rs:236-244)", in->format);
    if (out->format != in->format)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "colorlut: input and output formats differ");
    if (int rc = check_packed_frame(in, "colorlut input"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(out, "colorlut output"); rc != MVFX_OK) return rc;
    if (in->width != out->width || in->height != out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "colorlut: input %ux%u and output %ux%u differ", in->width, in->height, out->width, out->height);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (in->width == 0 || in->height == 0)
        return MVFX_OK;
    if (int rc = ensure_uploaded(h); rc != MVFX_OK) return rc;

    const CubeLut &l = h->lut;
    LutParams p{};
    p.cube = reinterpret_cast<const float4 *>(h->d_rgba);
    for (int c = 0; c < 3; c++) {
        p.t[c] = h->d_table[c];
        p.scale[c] = l.domain_scale[c];
        p.offset[c] = l.domain_offset[c];
    }
    p.size = l.size;
    p.size_m1 = (float)l.size - 1.0f;

    if (in->format == MVFX_FORMAT_RGB10A2_LE) {
        FrameBatch ifb10{}, ofb10{};
        uint64_t bits = (uint64_t)in->stride | out->stride;
        for (uint32_t i = 0; i < n; i++) {
            ifb10.base[i] = static_cast<uint8_t *>(ins[i].data);
            ofb10.base[i] = static_cast<uint8_t *>(outs[i].data);
            bits |= (uint64_t)(uintptr_t)ins[i].data | (uint64_t)(uintptr_t)outs[i].data;
        }
        if (bits & 3)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: RGB10A2_LE frames must be 4-byte aligned");
        const dim3 g10((in->width + kBlock - 1) / kBlock, in->height < 65535u ? in->height : 65535u, n);
        if (l.is_3d)
            MVFX_LAUNCH(colorlut_rgb10a2_kernel<true>, g10, dim3(kBlock), 0, st, ifb10, ofb10, in->width, in->height, (uint64_t)in->stride,
                               (uint64_t)out->stride, p);
        else
            MVFX_LAUNCH(colorlut_rgb10a2_kernel<false>, g10, dim3(kBlock), 0, st, ifb10, ofb10, in->width, in->height, (uint64_t)in->stride,
                               (uint64_t)out->stride, p);
        MVFX_HIP_TRY(hipGetLastError());
        return MVFX_OK;
    }

    const bool wide = in->format != MVFX_FORMAT_RGBA;
    const bool le = in->format != MVFX_FORMAT_RGBA64_BE;
    const uint32_t bpp = wide ? 8 : 4, pxv = wide ? 2 : 4;
    const bool flat = (uint64_t)in->width * bpp == in->stride && (uint64_t)out->width * bpp == out->stride;
    uint64_t width = in->width, is = in->stride, os = out->stride;
    uint32_t rows = in->height;
    FrameBatch ifb{}, ofb{};
    uint64_t align_or = 0;
    for (uint32_t i = 0; i < n; i++) {
        ifb.base[i] = static_cast<uint8_t *>(ins[i].data);
        ofb.base[i] = static_cast<uint8_t *>(outs[i].data);
        align_or |= (uint64_t)(uintptr_t)ins[i].data | (uint64_t)(uintptr_t)outs[i].data;
    }
    if (flat) { width = (uint64_t)in->width * in->height; rows = 1; is = os = 0; }
    else align_or |= is | os;
    const bool vec = (align_or & 15) == 0;

    if (!baking && !wide && opt_lut_placement() == 6) {
        if (!vec || (!flat && (in->width & 3) != 0) || (flat && (width & 3) != 0))
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: the baked table kernel needs 16-byte aligned rows and a width that is a multiple of 4");
        if (int rc = ensure_baked(h, st); rc != MVFX_OK) return rc;
        const uint64_t vecs = width / 4;
        constexpr int kPerLane = 2;
        const uint64_t bx = (vecs + 256u * kPerLane - 1) / (256u * kPerLane);
        if (bx > 0x7fffffffull) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: frame too large");
        MVFX_LAUNCH(colorlut_baked_kernel<kPerLane>, dim3((uint32_t)bx, rows < 65535u ? rows : 65535u, n), dim3(256), 0, st, ifb, ofb, vecs, rows,
                           is, os, h->d_baked);
        MVFX_HIP_TRY(hipGetLastError());
        return MVFX_OK;
    }

    const bool fits_lds = l.is_3d ? l.size <= kLds3dMaxSize : l.size <= kLds1dMaxSize;
    bool finite = true;
    for (int c = 0; c < 3; c++)
        finite = finite && std::isfinite(l.domain_scale[c]) && std::isfinite(l.domain_offset[c]);
    // 0 auto | 1 node layout in global/L2 | 2 LDS | 3 cell-packed global, per-lane gathers | 4 literal kernels |
    // 5 cell-packed global + wave-local 3x3x3 cell neighbourhood in LDS (16 x 16 pixel tiles)
    bool use_lds = fits_lds, use_cells = false, use_fast = finite && vec, use_tiles = false;
    const bool tiles_ok = l.is_3d && h->d_cells != nullptr && h->d_tile_tables != nullptr && finite && (in->width & 3) == 0 &&
                          ((align_or | in->stride | out->stride) & 15) == 0 && (uint64_t)in->stride * in->height < (1ull << 32) &&
                          (uint64_t)out->stride * out->height < (1ull << 32) && (in->height + 15) / 16 <= 65535u;
    switch (opt_lut_placement()) {
    case 1: use_lds = false; break;
    case 2:
        if (!fits_lds) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: LUT of size %u does not fit in LDS", l.size);
        use_lds = true; break;
    case 3:
        if (!h->d_cells) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: no cell-packed copy for this LUT (1-D or size > %u)", kCellMaxSize);
        use_lds = false; use_cells = true; break;
    case 4: use_fast = false; break;
    case 5:
        if (!tiles_ok) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: the tile kernel needs a 3-D LUT of size 3..%u, width %% 4 == 0 and 16-byte aligned rows", kCellMaxSize);
        use_lds = false; use_cells = true; use_tiles = true; break;
    default:
        // the tile kernel also beats the whole-cube-in-LDS kernel on cubes that fit LDS (17^3, 4K natural-like frame: 23.9 vs
        // 31.2 us per single-frame launch, 49.3 k vs 39.8 k fps with 16 frames per launch; flat bars 28.2 vs 30.5 us); only
        // on uniform-random colours is the LDS cube faster (44.7 vs 79.5 us) -- pictures are not that, and 33^3 has no LDS
        // alternative anyway.  The LDS cube stays for frames the tile kernel does not take (odd widths, unaligned rows,
        // RGBA64) and behind MVFX_OPT_LUT_PLACEMENT = 2.
        use_tiles = tiles_ok;
        if (use_tiles) use_lds = false;
        use_cells = !use_lds && h->d_cells != nullptr;
        break;
    }
    if (!use_fast) use_cells = false; // the literal kernels read the node layout
    const size_t lds_bytes = l.is_3d ? (size_t)l.size * l.size * l.size * 16 : (size_t)l.size * 12;
    p.cells = reinterpret_cast<const float4 *>(h->d_cells);
    p.tile_tables = h->d_tile_tables;
    p.xtable = reinterpret_cast<const float4 *>(h->d_xtable);
    p.xcoord = reinterpret_cast<const uint2 *>(h->d_xcoord);
    p.xcoord_wg = reinterpret_cast<const uint2 *>(h->d_xcoord_wg);
    p.fast.c_hi = wide ? 1.0f / 65535.0f : 1.0f / 255.0f;
    p.fast.c_lo = (float)((wide ? 1.0 / 65535.0 : 1.0 / 255.0) - (double)p.fast.c_hi);
    p.fast.out_scale = wide ? 65535.0f : 255.0f;
    p.fast.pred_half = std::nextafterf(0.5f, 0.0f);

    const uint64_t work = vec ? (width + pxv - 1) / pxv : width;
    dim3 grid;
    if (use_lds) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        uint64_t bx = (work + kLdsBlock - 1) / kLdsBlock;
        if (bx > (uint64_t)cus) bx = (uint64_t)cus; // persistent: one workgroup per CU pays the LDS fill once
        if (n > 1) { // the CUs are shared by the frames of the batch
            bx = (bx + n - 1) / n;
            if (bx == 0) bx = 1;
        }
        grid = dim3((uint32_t)bx, 1, n);
        if (rows > 1) { // row-structured frame: spread workgroups over rows instead
            uint32_t by = rows < (uint32_t)cus ? rows : (uint32_t)cus;
            if (n > 1) by = (by + n - 1) / n;
            grid = dim3(1, by ? by : 1, n);
        }
    } else {
        uint64_t bx = (work + kBlock - 1) / kBlock;
        if (bx > 65535u * 16u) bx = 65535u * 16u;
        grid = dim3((uint32_t)bx, rows < 65535u ? rows : 65535u, n);
    }
    const FrameBatch &ip = ifb, &op = ofb;

    if (use_fast && use_tiles) {
        // wave block: 64 x 16 pixels for RGBA8 on cubes up to 48 points per axis, 32 x 16 for bigger cubes (finer cells: the colours of a
        // smaller block stay in the window more often) and for RGBA64.  4K, 33^3 natural-like, 16 frames per launch / one frame / flat
        // bars one frame / 65^3 one frame / RGBA64 one frame:
        //   16 x 16 (lanes 4 x 16, 1 row each)  47.0 k fps  24.9 us  29.5 us  38.1 us  33.2 us      (round 2's first version)
        //   64 x 16 (16 x 4, 4 rows)            61.6 k      22.5     24.3     39.4     34.2
        //   32 x 16 (8 x 8, 2 rows)             57.0 k      21.5     28.1     37.8     32.4
        //   64 x 32 (16 x 4, 8 rows) 60.2 k / 31.0 us;  32 x 32 59.1 k / 23.9;  64 x 8 56.3 k / 22.4;  32 x 64 47.4 k / 34.3
        // RGBA8 on cubes of 4+ points: the x-prelerped kernel (placement 5 keeps the kernel below for A/B runs)
        // RGBA8 on cubes of 5+ points: the workgroup-window kernel (round 5); placement 7 keeps round 4's per-wave windows for A/B runs
        // Which of the two: by the content probe's last verdict (colorlut_probe_kernel) -- busy or no verdict yet: the workgroup window.
        // MVFX_XWG (environment, read once; experiments): 1 always the workgroup window, 0 never.
        bool wg_window = !wide && h->d_xtable && h->d_xcoord_wg && opt_lut_placement() != 5 && opt_lut_placement() != 7;
        if (wg_window) {
            static const int forced = [] { const char *e = std::getenv("MVFX_XWG"); return e ? std::atoi(e) : -1; }();
            if (forced >= 0) {
                wg_window = forced != 0;
            } else if (thread_options() & MVFX_OPT_LUT_WG_WINDOW) {
                // asked for by the caller
            } else if (h->h_probe) {
                if (h->probe_calls.fetch_add(1, std::memory_order_relaxed) % kProbeEvery == 0) // (not MVFX_LAUNCH: the probe is no part of the frame's work)
                    hipLaunchKernelGGL(colorlut_probe_kernel, dim3(1), dim3(256), 0, st, ifb.base[0], in->width, in->height, in->stride, h->h_probe);
                wg_window = __atomic_load_n(&h->h_probe[0], __ATOMIC_RELAXED) != 1u;
            }
        }
        if (wg_window) {
            const uint32_t tx_ = (in->width + 127) / 128, ty_ = (in->height + 8 * MVFX_XTILE_ROWS - 1) / (8 * MVFX_XTILE_ROWS);
            MVFX_LAUNCH(colorlut_xwg_kernel, dim3(tx_, ty_, n), dim3(kBlock), 0, st, ip, op, in->width, in->height, in->stride, out->stride, p);
            MVFX_HIP_TRY(hipGetLastError());
            return MVFX_OK;
        }
        if (!wide && h->d_xtable && opt_lut_placement() != 5) {
            const uint32_t tx_ = (in->width + 63) / 64, ty_ = (in->height + 4 * MVFX_XTILE_ROWS - 1) / (4 * MVFX_XTILE_ROWS);
            const dim3 xgrid((tx_ + kBlock / 64 - 1) / (kBlock / 64), ty_, n);
            MVFX_LAUNCH(colorlut_xtile_kernel<MVFX_XTILE_RW>, xgrid, dim3(kBlock), 0, st, ip, op, in->width, in->height, in->stride, out->stride, p);
            MVFX_HIP_TRY(hipGetLastError());
            return MVFX_OK;
        }
        const bool wide_block = !wide && l.size < 49;
        const uint32_t tile_w = wide_block ? 64 : 32, tile_h = 16;
        const uint32_t tiles_x = (in->width + tile_w - 1) / tile_w, tiles_y = (in->height + tile_h - 1) / tile_h;
        const dim3 tgrid((tiles_x + kBlock / 64 - 1) / (kBlock / 64), tiles_y, n);
#define MVFX_TK(WIDE, LE, A, R) MVFX_LAUNCH((colorlut_tile_kernel<WIDE, LE, A, R>), tgrid, dim3(kBlock), 0, st, ip, op, in->width, in->height, in->stride, out->stride, p)
        if (wide_block) MVFX_TK(false, true, 16, 4);
        else if (!wide) MVFX_TK(false, true, 8, 2);
        else if (le) MVFX_TK(true, true, 16, 4);   // RGBA64: 16 lanes x 2 pixels = the same 32 x 16 block
        else MVFX_TK(true, false, 16, 4);
#undef MVFX_TK
        MVFX_HIP_TRY(hipGetLastError());
        return MVFX_OK;
    }
    if (use_fast) {
#define MVFX_FG(IS3D, CELLS, WIDE, LE) \
    do { MVFX_LAUNCH((colorlut_fast_global_kernel<IS3D, CELLS, WIDE, LE>), grid, dim3(kBlock), 0, st, ip, op, width, rows, is, os, p); \
         MVFX_HIP_TRY(hipGetLastError()); return MVFX_OK; } while (0)
#define MVFX_FL(IS3D, WIDE, LE) \
    do { auto k = colorlut_fast_lds_kernel<IS3D, WIDE, LE>; \
         MVFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
         MVFX_LAUNCH(k, grid, dim3(kLdsBlock), lds_bytes, st, ip, op, width, rows, is, os, p); \
         MVFX_HIP_TRY(hipGetLastError()); return MVFX_OK; } while (0)
        if (use_lds) {
            if (l.is_3d) { if (!wide) MVFX_FL(true, false, true); else if (le) MVFX_FL(true, true, true); else MVFX_FL(true, true, false); }
            else { if (!wide) MVFX_FL(false, false, true); else if (le) MVFX_FL(false, true, true); else MVFX_FL(false, true, false); }
        } else if (l.is_3d && use_cells) {
            if (!wide) MVFX_FG(true, true, false, true); else if (le) MVFX_FG(true, true, true, true); else MVFX_FG(true, true, true, false);
        } else if (l.is_3d) {
            if (!wide) MVFX_FG(true, false, false, true); else if (le) MVFX_FG(true, false, true, true); else MVFX_FG(true, false, true, false);
        } else {
            if (!wide) MVFX_FG(false, false, false, true); else if (le) MVFX_FG(false, false, true, true); else MVFX_FG(false, false, true, false);
        }
#undef MVFX_FG
#undef MVFX_FL
    }

#define MVFX_GO(IS3D, WIDE, LE, VEC) \
    return launch_one<IS3D, WIDE, LE, VEC>(use_lds, grid, lds_bytes, st, ip, op, width, rows, is, os, p)
    if (l.is_3d) {
        if (!wide) { if (vec) MVFX_GO(true, false, true, true); else MVFX_GO(true, false, true, false); }
        else if (le) { if (vec) MVFX_GO(true, true, true, true); else MVFX_GO(true, true, true, false); }
        else { if (vec) MVFX_GO(true, true, false, true); else MVFX_GO(true, true, false, false); }
    } else {
        if (!wide) { if (vec) MVFX_GO(false, false, true, true); else MVFX_GO(false, false, true, false); }
        else if (le) { if (vec) MVFX_GO(false, true, true, true); else MVFX_GO(false, true, true, false); }
        else { if (vec) MVFX_GO(false, true, false, true); else MVFX_GO(false, true, false, false); }
    }
#undef MVFX_GO
}


// colorlut on an I420 frame: the fused kernel when the fast LUT path and the tile layout apply, else the same three
// steps through two RGBA scratch frames (identical bytes either way).
int colorlut_i420_impl(mvfx_cube_lut *h, const mvfx_planar_frame *in, const mvfx_planar_frame *out, int yuv_standard, hipStream_t st)
{
    if (!h)
        return fail(MVFX_ERR_NO_LUT, "colorlut: No LUT configured (colorlut/imp.rs:209-213)");
    if (!in || !out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: NULL frame");
    if (in->format != MVFX_FORMAT_I420 || out->format != MVFX_FORMAT_I420)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "colorlut_i420: both frames must be I420");
    if (in->width != out->width || in->height != out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "colorlut: input %ux%u and output %ux%u differ", in->width, in->height, out->width, out->height);
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut_i420: yuv_standard %d is not 0..3", yuv_standard);
    const uint32_t w = in->width, hgt = in->height;
    if ((w & 1) || (hgt & 1))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut_i420: odd-sized frame %ux%u (RGBA -> I420 needs even sizes)", w, hgt);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (w == 0 || hgt == 0) return MVFX_OK;
    const uint32_t cw = w / 2;
    for (int pidx = 0; pidx < 3; pidx++) {
        const uint32_t need = pidx == 0 ? w : cw;
        if (!in->data[pidx] || !out->data[pidx] || in->stride[pidx] < need || out->stride[pidx] < need)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut_i420: bad plane %d", pidx);
    }
    for (int pidx = 0; pidx < 3; pidx++)
        if (in->data[pidx] == out->data[pidx]) // colorlut is NeverInPlace (colorlut/imp.rs:162-166); the fused kernel also reads
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut_i420: input and output planes must not alias"); // neighbour input pixels
    if (int rc = ensure_uploaded(h); rc != MVFX_OK) return rc;
    const CubeLut &l = h->lut;
    bool finite = true;
    for (int c = 0; c < 3; c++)
        finite = finite && std::isfinite(l.domain_scale[c]) && std::isfinite(l.domain_offset[c]);
    uint64_t bits = 0;
    for (int pidx = 0; pidx < 3; pidx++) {
        const uint64_t v = (uint64_t)(uintptr_t)in->data[pidx] | in->stride[pidx] | (uint64_t)(uintptr_t)out->data[pidx] | out->stride[pidx];
        bits |= pidx == 0 ? (v & 7) : (v & 3);
    }
    const bool fused = finite && bits == 0 && (w % 8) == 0 && hgt / 2 <= 65535u && opt_lut_placement() != 4;
    if (fused) {
        LutParams p{};
        p.cube = reinterpret_cast<const float4 *>(h->d_rgba);
        p.cells = reinterpret_cast<const float4 *>(h->d_cells);
        for (int c = 0; c < 3; c++) { p.t[c] = h->d_table[c]; p.scale[c] = l.domain_scale[c]; p.offset[c] = l.domain_offset[c]; }
        p.size = l.size;
        p.size_m1 = (float)l.size - 1.0f;
        p.fast.c_hi = 1.0f / 255.0f;
        p.fast.c_lo = (float)(1.0 / 255.0 - (double)p.fast.c_hi);
        p.fast.out_scale = 255.0f;
        p.fast.pred_half = std::nextafterf(0.5f, 0.0f);
        const I420Planes pl{static_cast<const uint8_t *>(in->data[0]), static_cast<const uint8_t *>(in->data[1]), static_cast<const uint8_t *>(in->data[2]),
                            static_cast<uint8_t *>(out->data[0]), static_cast<uint8_t *>(out->data[1]), static_cast<uint8_t *>(out->data[2]),
                            in->stride[0], in->stride[1], in->stride[2], out->stride[0], out->stride[1], out->stride[2]};
        const int std_ = pick_yuv_standard(hgt, yuv_standard);
        const YuvToRgbCoef kin = yuv_to_rgb_coef(std_);
        const RgbToYuvCoef kout = rgb_to_yuv_coef(std_);
        const dim3 grid((w / 8 + kI420Block - 1) / kI420Block, hgt / 2);
        p.tile_tables = h->d_tile_tables;
        p.xtable = reinterpret_cast<const float4 *>(h->d_xtable);
        p.xcoord = reinterpret_cast<const uint2 *>(h->d_xcoord);
        p.xcoord_wg = reinterpret_cast<const uint2 *>(h->d_xcoord_wg);
    p.xcoord_wg = reinterpret_cast<const uint2 *>(h->d_xcoord_wg);
        if (l.is_3d && h->d_cells && h->d_tile_tables && hgt / 16 + 1 <= 65535u) {
            const dim3 tgrid((w + 255) / 256, (hgt + 15) / 16);
            if (h->d_xtable && opt_lut_placement() != 5)
                MVFX_LAUNCH(colorlut_i420_xtile_kernel, tgrid, dim3(kI420Block), 0, st, pl, w, hgt, p, kin, kout);
            else
                MVFX_LAUNCH(colorlut_i420_tile_kernel, tgrid, dim3(kI420Block), 0, st, pl, w, hgt, p, kin, kout);
        } else if (l.is_3d && h->d_cells)
            MVFX_LAUNCH((colorlut_i420_kernel<true, true>), grid, dim3(kI420Block), 0, st, pl, w, hgt, p, kin, kout);
        else if (l.is_3d)
            MVFX_LAUNCH((colorlut_i420_kernel<true, false>), grid, dim3(kI420Block), 0, st, pl, w, hgt, p, kin, kout);
        else
            MVFX_LAUNCH((colorlut_i420_kernel<false, false>), grid, dim3(kI420Block), 0, st, pl, w, hgt, p, kin, kout);
        MVFX_HIP_TRY(hipGetLastError());
        return MVFX_OK;
    }
    // three steps through scratch
    void *ra = nullptr, *rb = nullptr;
    const size_t bytes = (size_t)w * 4 * hgt;
    if (int rc = host_scratch(bytes, 0, &ra); rc != MVFX_OK) return rc;
    if (int rc = host_scratch(bytes, 1, &rb); rc != MVFX_OK) return rc;
    mvfx_frame fa{ra, w, hgt, w * 4, MVFX_FORMAT_RGBA}, fb{rb, w, hgt, w * 4, MVFX_FORMAT_RGBA};
    if (int rc = mvfx_convert_i420_to_rgba(in, &fa, yuv_standard, reinterpret_cast<mvfx_stream>(st)); rc != MVFX_OK) return rc;
    if (int rc = colorlut_impl(h, &fa, &fb, 1, st); rc != MVFX_OK) return rc;
    return mvfx_convert_rgba_to_i420(&fb, out, yuv_standard, reinterpret_cast<mvfx_stream>(st));
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_cube_lut_parse(const char *text, size_t len, mvfx_cube_lut **out)
{
    if (!out || (!text && len))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "cube_lut_parse: NULL argument");
    *out = nullptr;
    auto *h = new (std::nothrow) mvfx_cube_lut();
    if (!h)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "cube_lut_parse: out of memory");
    std::string err;
    if (!parse_cube(std::string_view(text ? text : "", len), h->lut, err)) {
        delete h;
        return fail(MVFX_ERR_PARSE, "Invalid LUT: %s", err.c_str());
    }
    *out = h;
    return MVFX_OK;
}

int mvfx_cube_lut_parse_file(const char *path, mvfx_cube_lut **out)
{
    if (!out || !path)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "cube_lut_parse_file: NULL argument");
    *out = nullptr;
    auto *h = new (std::nothrow) mvfx_cube_lut();
    if (!h)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "cube_lut_parse_file: out of memory");
    std::string err;
    bool io = false;
    if (!parse_cube_file(path, h->lut, err, io)) {
        delete h;
        return fail(io ? MVFX_ERR_IO : MVFX_ERR_PARSE, "Failed to parse LUT file %s: %s", path, err.c_str());
    }
    *out = h;
    return MVFX_OK;
}

void mvfx_cube_lut_free(mvfx_cube_lut *lut)
{
    if (!lut) return;
    if (lut->d_rgba) (void)hipFree(lut->d_rgba);
    if (lut->d_cells) (void)hipFree(lut->d_cells);
    if (lut->d_xtable) (void)hipFree(lut->d_xtable);
    if (lut->d_xcoord) (void)hipFree(lut->d_xcoord);
    if (lut->d_xcoord_wg) (void)hipFree(lut->d_xcoord_wg);
    if (lut->h_probe) (void)hipHostFree(lut->h_probe);
    if (lut->d_tile_tables) (void)hipFree(lut->d_tile_tables);
    if (lut->d_baked) (void)hipFree(lut->d_baked);
    for (auto &t : lut->d_table) if (t) (void)hipFree(t);
    delete lut;
}

int mvfx_cube_lut_write(const mvfx_cube_lut *lut, char **text_out, size_t *len_out)
{
    if (!lut || !text_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "cube_lut_write: NULL argument");
    const std::string text = write_cube(lut->lut);
    char *buf = static_cast<char *>(malloc(text.size() + 1));
    if (!buf)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "cube_lut_write: %zu bytes", text.size() + 1);
    std::memcpy(buf, text.c_str(), text.size() + 1);
    *text_out = buf;
    if (len_out) *len_out = text.size();
    return MVFX_OK;
}

void mvfx_free_text(char *text) { free(text); }

int mvfx_cube_lut_is_3d(const mvfx_cube_lut *lut) { return lut && lut->lut.is_3d ? 1 : 0; }
uint32_t mvfx_cube_lut_size(const mvfx_cube_lut *lut) { return lut ? lut->lut.size : 0; }
int mvfx_cube_lut_content_verdict(const mvfx_cube_lut *lut, uint32_t *busy_blocks)
{
    if (!lut || !lut->h_probe) {
        if (busy_blocks) *busy_blocks = 0;
        return 0;
    }
    if (busy_blocks) *busy_blocks = __atomic_load_n(&lut->h_probe[1], __ATOMIC_RELAXED);
    return (int)__atomic_load_n(&lut->h_probe[0], __ATOMIC_RELAXED);
}

int mvfx_cube_lut_domain(const mvfx_cube_lut *lut, float scale[3], float offset[3])
{
    if (!lut || !scale || !offset)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "cube_lut_domain: NULL argument");
    std::memcpy(scale, lut->lut.domain_scale, sizeof(float) * 3);
    std::memcpy(offset, lut->lut.domain_offset, sizeof(float) * 3);
    return MVFX_OK;
}

const float *mvfx_cube_lut_rgba(const mvfx_cube_lut *lut)
{
    return lut && lut->lut.is_3d ? lut->lut.rgba.data() : nullptr;
}

const float *mvfx_cube_lut_table_1d(const mvfx_cube_lut *lut, int channel)
{
    if (!lut || lut->lut.is_3d || channel < 0 || channel > 2) return nullptr;
    return lut->lut.table[channel].data();
}

int mvfx_colorlut_transform_frame(mvfx_cube_lut *lut, const mvfx_frame *in_frame, const mvfx_frame *out_frame,
                                  mvfx_stream stream)
{
    return colorlut_impl(lut, in_frame, out_frame, 1, as_stream(stream));
}

int mvfx_colorlut_transform_frames(mvfx_cube_lut *lut, const mvfx_frame *in_frames, const mvfx_frame *out_frames,
                                   uint32_t n_frames, mvfx_stream stream)
{
    return colorlut_impl(lut, in_frames, out_frames, n_frames, as_stream(stream));
}

int mvfx_colorlut_transform_i420(mvfx_cube_lut *lut, const mvfx_planar_frame *i420_in, const mvfx_planar_frame *i420_out,
                                 int32_t yuv_standard, mvfx_stream stream)
{
    return colorlut_i420_impl(lut, i420_in, i420_out, yuv_standard, as_stream(stream));
}

int mvfx_colorlut_transform_frame_host(mvfx_cube_lut *lut, const mvfx_frame *in_frame, const mvfx_frame *out_frame)
{
    if (!lut)
        return fail(MVFX_ERR_NO_LUT, "colorlut: No LUT configured (colorlut/imp.rs:209-213)");
    if (!in_frame || !out_frame)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: NULL frame");
    if (int rc = check_packed_frame(in_frame, "colorlut input"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(out_frame, "colorlut output"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t ib = (size_t)in_frame->stride * in_frame->height;
    const size_t ob = (size_t)out_frame->stride * out_frame->height;
    if (ib == 0 || ob == 0)
        return colorlut_impl(lut, in_frame, out_frame, 1, nullptr);
    void *din = nullptr, *dout = nullptr;
    if (int rc = host_scratch(ib, 0, &din); rc != MVFX_OK) return rc;
    if (int rc = host_scratch(ob, 1, &dout); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    MVFX_HIP_TRY(hipMemcpyAsync(din, in_frame->data, ib, hipMemcpyHostToDevice, st));
    const uint32_t bpp = (uint32_t)bytes_per_pixel(out_frame->format);
    if ((size_t)out_frame->width * bpp != out_frame->stride) // keep the caller's row padding bytes
        MVFX_HIP_TRY(hipMemcpyAsync(dout, out_frame->data, ob, hipMemcpyHostToDevice, st));
    mvfx_frame di = *in_frame, dof = *out_frame;
    di.data = din;
    dof.data = dout;
    if (int rc = colorlut_impl(lut, &di, &dof, 1, st); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(out_frame->data, dout, ob, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return MVFX_OK;
}

} // extern "C"
