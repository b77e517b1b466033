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
#include "colorlut_device.hpp"
#include "direct_dispatch_colorlut.h"

#include "cube_parser.h"
#include "device_replicas.h"

#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

// The device side of a LUT on ONE device: made by the first transform on that device (ensure_uploaded), kept until the handle is freed.
// One process can drive several GPUs -- streaming threads follow the device of their input memory, the reference's d3d12colorlut does the
// same with its context (video/colorlut/src/d3d12colorlut/imp.rs:494-542) --, so a handle keeps one replica PER device instead of freeing
// and re-uploading on a switch (round 5 held a single copy).
struct LutDeviceCopy {
    std::mutex mu;         // serialises the upload on this device
    bool ready = false;
    float *d_rgba = nullptr;
    uint32_t *d_tile_tables = nullptr; // tile kernel: 3 x 256 x (cell index, fraction) per byte value + 192 neighbourhood piece offsets
    uint32_t *d_xcoord = nullptr; // colorlut_xtile_kernel: per byte value of g and b {cell index x LDS row pitch, fraction bits}
    uint32_t *d_xcoord_wg = nullptr; // colorlut_xwg_kernel: the same with its window's pitches (cubes of 5+ points)
    // Content probe (round 5, see colorlut_probe_kernel): which of the two window kernels the automatic choice takes for this LUT's frames.
    // The probe kernel writes its verdict into a page-locked host word; the launcher reads it without synchronising (a verdict a few
    // launches old is as good: pictures of a stream resemble their predecessors) -- advisory state, both kernels produce the same bytes.
    uint32_t *h_probe = nullptr;            // [0]: 0 = no verdict yet, 1 = calm, 2 = busy; [1]: busy blocks of the last probe (of 256)
    std::atomic<uint32_t> probe_calls{0};
    std::atomic<uint64_t> probe_geom{0};    // width << 32 | height of the last frame the automatic choice saw
    float *d_xtable = nullptr; // x-prelerped table of colorlut_xtile_kernel: [y][z][r byte] x (X.rgb, D.rgb) f32 = 24 B (3-D, 4 <= size <= kCellMaxSize)
    float *d_cells = nullptr; // cell-packed copy: size^3 cells x 8 corners x (r,g,b) f32 = 96 B (3-D, size <= kCellMaxSize)
    float *d_table[3] = {nullptr, nullptr, nullptr};
    // baked table (placement 6): the LUT applied to every one of the 2^24 RGB byte triples, 64 MiB, entry = output R | G << 8 | B << 16
    // at index r | g << 8 | b << 16 -- produced by running this file's own interpolating kernels once over a 4096 x 4096 frame that
    // holds every colour, so its bytes are theirs by construction
    std::mutex bake_mu;
    uint32_t *d_baked = nullptr;
};

struct mvfx_cube_lut {
    mvfx::CubeLut lut;
    mvfx::DeviceReplicas<LutDeviceCopy> copies; // by device ordinal
};

namespace mvfx {
namespace {

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


void free_device_copy(LutDeviceCopy &d) // on the device that is current
{
    if (d.d_rgba) { (void)hipFree(d.d_rgba); d.d_rgba = nullptr; }
    if (d.d_cells) { (void)hipFree(d.d_cells); d.d_cells = nullptr; }
    if (d.d_xtable) { (void)hipFree(d.d_xtable); d.d_xtable = nullptr; }
    if (d.d_xcoord) { (void)hipFree(d.d_xcoord); d.d_xcoord = nullptr; }
    if (d.d_xcoord_wg) { (void)hipFree(d.d_xcoord_wg); d.d_xcoord_wg = nullptr; }
    if (d.h_probe) { (void)hipHostFree(d.h_probe); d.h_probe = nullptr; }
    if (d.d_tile_tables) { (void)hipFree(d.d_tile_tables); d.d_tile_tables = nullptr; }
    if (d.d_baked) { (void)hipFree(d.d_baked); d.d_baked = nullptr; }
    for (auto &t : d.d_table) if (t) { (void)hipFree(t); t = nullptr; }
    d.ready = false;
}

// The replica of the calling thread's current device, uploaded on its first use there (every user of the handle on one device shares it).
int ensure_uploaded(mvfx_cube_lut *h, LutDeviceCopy **out)
{
    int dev = 0;
    MVFX_HIP_TRY(hipGetDevice(&dev));
    LutDeviceCopy *d = h->copies.get_or_create(dev, [] { return new (std::nothrow) LutDeviceCopy(); });
    if (!d)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "colorlut: no replica slot for device %d (ordinals 0..%d)", dev, h->copies.capacity() - 1);
    *out = d;
    std::lock_guard<std::mutex> lock(d->mu);
    if (d->ready)
        return MVFX_OK;
    free_device_copy(*d); // the remains of an upload that failed half way
    const CubeLut &l = h->lut;
    if (l.is_3d) {
        const size_t bytes = l.rgba.size() * sizeof(float);
        MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_rgba), bytes));
        MVFX_HIP_TRY(hipMemcpy(d->d_rgba, l.rgba.data(), bytes, hipMemcpyHostToDevice));
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
            MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_cells), cells.size() * sizeof(float)));
            MVFX_HIP_TRY(hipMemcpy(d->d_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
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
                MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_tile_tables), tt.size() * sizeof(uint32_t)));
                MVFX_HIP_TRY(hipMemcpy(d->d_tile_tables, tt.data(), tt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                if (l.size >= 4) { // the x-prelerped table of colorlut_xtile_kernel, computed on the device from the two copies above
                    const size_t entries = (size_t)l.size * (l.size + 1) * 256;
                    MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_xtable), entries * 6 * sizeof(float)));
                    launch_colorlut_xtable_build(reinterpret_cast<const float4 *>(d->d_rgba), d->d_tile_tables, l.size, d->d_xtable, entries);
                    MVFX_HIP_TRY(hipGetLastError());
                    std::vector<uint32_t> xc(512 * 2);
                    for (int b = 0; b < 256; b++) {
                        xc[2 * b] = tt[(256 + b) * 2] * kXPitchY;          xc[2 * b + 1] = tt[(256 + b) * 2 + 1];
                        xc[2 * (256 + b)] = tt[(512 + b) * 2] * kXPitchZ;  xc[2 * (256 + b) + 1] = tt[(512 + b) * 2 + 1];
                    }
                    MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_xcoord), xc.size() * sizeof(uint32_t)));
                    MVFX_HIP_TRY(hipMemcpy(d->d_xcoord, xc.data(), xc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    if (l.size >= kWgNY && l.size >= kWgNZ) { // colorlut_xwg_kernel's window is 5 x 5 cells
                        for (int b = 0; b < 256; b++) {
                            xc[2 * b] = tt[(256 + b) * 2] * kWgPitchY;
                            xc[2 * (256 + b)] = tt[(512 + b) * 2] * kWgPitchZ;
                        }
                        MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_xcoord_wg), xc.size() * sizeof(uint32_t)));
                        MVFX_HIP_TRY(hipMemcpy(d->d_xcoord_wg, xc.data(), xc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                        if (!d->h_probe) { // (without it the automatic choice is always the workgroup-window kernel)
                            void *q = nullptr;
                            if (hipHostMalloc(&q, 64, hipHostMallocDefault) == hipSuccess) {
                                std::memset(q, 0, 64);
                                d->h_probe = static_cast<uint32_t *>(q);
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
            MVFX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->d_table[c]), bytes));
            MVFX_HIP_TRY(hipMemcpy(d->d_table[c], l.table[c].data(), bytes, hipMemcpyHostToDevice));
        }
    }
    d->ready = true;
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
int ensure_baked(mvfx_cube_lut *h, LutDeviceCopy *d, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(d->bake_mu);
    if (__atomic_load_n(&d->d_baked, __ATOMIC_ACQUIRE)) return MVFX_OK;
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
    __atomic_store_n(&d->d_baked, table, __ATOMIC_RELEASE);
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
    LutDeviceCopy *d = nullptr; // this device's replica
    if (int rc = ensure_uploaded(h, &d); rc != MVFX_OK) return rc;

    const CubeLut &l = h->lut;
    LutParams p{};
    p.cube = reinterpret_cast<const float4 *>(d->d_rgba);
    for (int c = 0; c < 3; c++) {
        p.t[c] = d->d_table[c];
        p.scale[c] = l.domain_scale[c];
        p.offset[c] = l.domain_offset[c];
    }
    p.size = l.size;
    p.size_m1 = (float)l.size - 1.0f;

    // MVFX_OPT_DIRECT_ONLY (the caller holds a direct fence open on a lane queue): only a launch the lane takes may happen
    const bool direct_only = opt_direct_only() && !baking;
    const auto no_lane = [] { return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "colorlut: the direct-dispatch lane does not take this call"); };
    if (in->format == MVFX_FORMAT_RGB10A2_LE) {
        if (direct_only) return no_lane();
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
        if (direct_only) return no_lane();
        if (!vec || (!flat && (in->width & 3) != 0) || (flat && (width & 3) != 0))
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: the baked table kernel needs 16-byte aligned rows and a width that is a multiple of 4");
        if (int rc = ensure_baked(h, d, st); rc != MVFX_OK) return rc;
        const uint64_t vecs = width / 4;
        constexpr int kPerLane = 2;
        const uint64_t bx = (vecs + 256u * kPerLane - 1) / (256u * kPerLane);
        if (bx > 0x7fffffffull) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: frame too large");
        MVFX_LAUNCH(colorlut_baked_kernel<kPerLane>, dim3((uint32_t)bx, rows < 65535u ? rows : 65535u, n), dim3(256), 0, st, ifb, ofb, vecs, rows,
                           is, os, d->d_baked);
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
    const bool tiles_ok = l.is_3d && d->d_cells != nullptr && d->d_tile_tables != nullptr && finite && (in->width & 3) == 0 &&
                          ((align_or | in->stride | out->stride) & 15) == 0 && (uint64_t)in->stride * in->height < (1ull << 32) &&
                          (uint64_t)out->stride * out->height < (1ull << 32) && (in->height + 15) / 16 <= 65535u;
    switch (opt_lut_placement()) {
    case 1: use_lds = false; break;
    case 2:
        if (!fits_lds) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: LUT of size %u does not fit in LDS", l.size);
        use_lds = true; break;
    case 3:
        if (!d->d_cells) return fail(MVFX_ERR_INVALID_ARGUMENT, "colorlut: no cell-packed copy for this LUT (1-D or size > %u)", kCellMaxSize);
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
        use_cells = !use_lds && d->d_cells != nullptr;
        break;
    }
    if (!use_fast) use_cells = false; // the literal kernels read the node layout
    // what the lane takes: one RGBA8 frame through one of the two x-prelerped window kernels (csrc/direct/colorlut_direct_kernels.hip)
    const bool lane_kernels = use_fast && use_tiles && !wide && n == 1 && d->d_xtable && opt_lut_placement() != 5;
    if (direct_only && !lane_kernels) return no_lane();
    const size_t lds_bytes = l.is_3d ? (size_t)l.size * l.size * l.size * 16 : (size_t)l.size * 12;
    p.cells = reinterpret_cast<const float4 *>(d->d_cells);
    p.tile_tables = d->d_tile_tables;
    p.xtable = reinterpret_cast<const float4 *>(d->d_xtable);
    p.xcoord = reinterpret_cast<const uint2 *>(d->d_xcoord);
    p.xcoord_wg = reinterpret_cast<const uint2 *>(d->d_xcoord_wg);
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
        bool wg_window = !wide && d->d_xtable && d->d_xcoord_wg && opt_lut_placement() != 5 && opt_lut_placement() != 7;
        if (wg_window) {
            static const int forced = [] { const char *e = std::getenv("MVFX_XWG"); return e ? std::atoi(e) : -1; }();
            if (forced >= 0) {
                wg_window = forced != 0;
            } else if (thread_options() & MVFX_OPT_LUT_WG_WINDOW) {
                // asked for by the caller
            } else if (d->h_probe) {
                // another geometry than the last call's is another stream (or a caps change): look at once instead of up to 31 launches later
                const uint64_t geom = ((uint64_t)in->width << 32) | in->height;
                if (d->probe_geom.exchange(geom, std::memory_order_relaxed) != geom) d->probe_calls.store(0, std::memory_order_relaxed);
                if (d->probe_calls.fetch_add(1, std::memory_order_relaxed) % kProbeEvery == 0)
                    launch_colorlut_probe(st, ifb.base[0], in->width, in->height, in->stride, d->h_probe);
                wg_window = __atomic_load_n(&d->h_probe[0], __ATOMIC_RELAXED) != 1u;
            }
        }
        if (lane_kernels && opt_direct() && !baking) {
            // The direct-dispatch lane: the same kernel bodies with write-through stores, as a packet of the library's own without a release fence
            // (direct_dispatch.h).  Needs the thread's completion event -- it becomes the frame's direct fence; the content probe above stays on `st`.
            DirectLutArgs da{};
            da.in = ifb.base[0];
            da.out = ofb.base[0];
            da.width = in->width; da.height = in->height; da.in_stride = in->stride; da.out_stride = out->stride;
            da.p = p;
            const int rc = direct_colorlut_submit(da, wg_window, direct_queue_hint(st), !opt_direct_unordered());
            if (rc == MVFX_OK) return MVFX_OK;
            if (rc < 0) return rc;
            if (direct_only) return no_lane();
        }
        if (wg_window) {
            const uint32_t tx_ = (in->width + 127) / 128, ty_ = (in->height + 8 * kXRows - 1) / (8 * kXRows);
            launch_colorlut_xwg(dim3(tx_, ty_, n), st, ip, op, in->width, in->height, in->stride, out->stride, p);
            MVFX_HIP_TRY(hipGetLastError());
            return MVFX_OK;
        }
        if (!wide && d->d_xtable && opt_lut_placement() != 5) {
            const uint32_t tx_ = (in->width + 63) / 64, ty_ = (in->height + 4 * kXRows - 1) / (4 * kXRows);
            const dim3 xgrid((tx_ + kBlock / 64 - 1) / (kBlock / 64), ty_, n);
            launch_colorlut_xtile(xgrid, st, ip, op, in->width, in->height, in->stride, out->stride, p);
            MVFX_HIP_TRY(hipGetLastError());
            return MVFX_OK;
        }
        const bool wide_block = !wide && l.size < 49;
        const uint32_t tile_w = wide_block ? 64 : 32, tile_h = 16;
        const uint32_t tiles_x = (in->width + tile_w - 1) / tile_w, tiles_y = (in->height + tile_h - 1) / tile_h;
        const dim3 tgrid((tiles_x + kBlock / 64 - 1) / (kBlock / 64), tiles_y, n);
        // RGBA64: 16 lanes x 2 pixels = the same 32 x 16 block
        launch_colorlut_tile(wide, le, !wide && !wide_block, tgrid, st, ip, op, in->width, in->height, in->stride, out->stride, p);
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
    LutDeviceCopy *d = nullptr; // this device's replica
    if (int rc = ensure_uploaded(h, &d); rc != MVFX_OK) return rc;
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
        p.cube = reinterpret_cast<const float4 *>(d->d_rgba);
        p.cells = reinterpret_cast<const float4 *>(d->d_cells);
        for (int c = 0; c < 3; c++) { p.t[c] = d->d_table[c]; p.scale[c] = l.domain_scale[c]; p.offset[c] = l.domain_offset[c]; }
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
        p.tile_tables = d->d_tile_tables;
        p.xtable = reinterpret_cast<const float4 *>(d->d_xtable);
        p.xcoord = reinterpret_cast<const uint2 *>(d->d_xcoord);
        p.xcoord_wg = reinterpret_cast<const uint2 *>(d->d_xcoord_wg);
        if (l.is_3d && d->d_cells && d->d_tile_tables && hgt / 16 + 1 <= 65535u) {
            const dim3 tgrid((w + 255) / 256, (hgt + 15) / 16);
            launch_colorlut_i420_window(d->d_xtable && opt_lut_placement() != 5, tgrid, st, pl, w, hgt, p, kin, kout);
        } else if (l.is_3d && d->d_cells)
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
    // every device's replica, on its device (hipFree of another device's pointer works, but the probe word and the order of frees are per device)
    int before = -1;
    const bool have_dev = hipGetDevice(&before) == hipSuccess;
    lut->copies.for_each([&](int dev, LutDeviceCopy &d) {
        if (have_dev && dev != before) (void)hipSetDevice(dev);
        mvfx::direct_quiesce(dev); // a lane kernel of the last frames may still be reading the tables: hipFree does not wait for queues of our own
        mvfx::free_device_copy(d);
    });
    if (have_dev) { int now = -1; if (hipGetDevice(&now) == hipSuccess && now != before) (void)hipSetDevice(before); }
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
int mvfx_cube_lut_device_copies(const mvfx_cube_lut *lut) { return lut ? lut->copies.count() : 0; }
int mvfx_cube_lut_content_verdict(const mvfx_cube_lut *lut, uint32_t *busy_blocks)
{
    int dev = -1; // the replica of the calling thread's device
    const LutDeviceCopy *d = lut && hipGetDevice(&dev) == hipSuccess ? lut->copies.find(dev) : nullptr;
    if (!d || !d->h_probe) {
        if (busy_blocks) *busy_blocks = 0;
        return 0;
    }
    if (busy_blocks) *busy_blocks = __atomic_load_n(&d->h_probe[1], __ATOMIC_RELAXED);
    return (int)__atomic_load_n(&d->h_probe[0], __ATOMIC_RELAXED);
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
