// hsvfilter / hsvdetector kernels for gfx950 and their C-ABI launchers.
//
// Replaces the per-pixel loops of
//   video/hsv/src/hsvfilter/imp.rs:76-120 (+ format dispatch :322-377)
//   video/hsv/src/hsvdetector/imp.rs:100-160 (+ 24 closure pairs :422-707)
//
// Memory plan (streaming, every pixel independent; LDS only holds the 8-entry sextant selector table):
//   4-byte formats: one lane owns 4 consecutive pixels = one 16-byte global_load_dwordx4 /
//   global_store_dwordx4, so a wave64 touches 1 KiB contiguous per instruction; in launches of several frames a
//   lane owns two such groups one workgroup-width apart and issues both loads before the arithmetic.  A frame
//   whose stride equals width*4 is treated as ONE row of width*height pixels (no per-row tail).
//   mvfx_thread_set_options(MVFX_OPT_NONTEMPORAL) adds the non-temporal hint to those loads/stores.
//   3-byte formats: one lane owns 4 pixels = 12 bytes = global_load_dwordx3, rows stay
//   dword-coalesced; the <4-pixel row tail is done bytewise by the owning lane.
//   Frames that are not 16-byte (4-byte formats) / 4-byte (3-byte formats) aligned fall back to
//   a dword-per-pixel or byte-per-channel kernel: slower, still on the GPU, same results.
//   Grid: x = pixel groups (grid-stride), y = rows, z = frame of the batch; >= 4K workgroups
//   for a 4K frame so all 256 CUs / 8 XCDs are covered many times over; consecutive
//   workgroups stream consecutive addresses so each XCD's L2 sees disjoint lines (no reuse to
//   exploit; an XCD-contiguous remap was measured 1 % slower).
//   This file is compiled with the ILP-driven scheduling strategy (Makefile): the pixel function is one long
//   dependent chain and the four pixels of a group have to be interleaved by the scheduler (+3.8 %).
#include "hsv_math.hpp"
#include "hsv_filter_lds.hpp"
#include "device_store.hpp"
#include "direct_dispatch.h"
#include "convert_math.hpp"
#include "mvfx_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace mvfx {
namespace {

constexpr int kBlock = 256; // 128 / 512 / 1024 measured slower (profiles/r1/ab_steady_block_nt.txt)
constexpr int kTile = 2;    // 16-byte pixel groups per lane in hsvfilter4_kernel (vec4 mode): both loads are issued first
                            // (1: -8 %, 3: -0.4 %, 4: -2 % under the ILP scheduling strategy, ab_steady_sched_strategy.txt).
                            // Launches smaller than one 4K frame keep one group per lane (more workgroups to balance).
                            // Round 1 kept one group per lane up to two 4K frames because an ISOLATED launch followed by a
                            // synchronisation finishes sooner that way (16.8 vs 18.4 us); what an element on device memory
                            // needs is throughput with launches of several streams in flight, and there two groups per lane
                            // win: 16 threads x single-frame launches 72.3 k -> 78.2 k frames/s, one thread 62.7 k -> 64.2 k
                            // (profiles/r2/streams_sweep_tile1.txt / _tile2.txt, same box, batch-16 launches 78.8 k).
constexpr uint64_t kTileMinGroups = 1ull * 3840 * 2160 / 4;

enum : int { kModeBytes = 0, kModeVec4 = 1, kModeDword = 2 };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- channel placement -------------------------------------------------------------------
// hsvfilter/imp.rs:327-373: OFF = index of the first colour byte, BGR = byte order of the triple
template <int OFF, bool BGR>
__device__ __forceinline__ void unpack4(uint32_t px, uint32_t &R, uint32_t &G, uint32_t &B)
{
    const uint32_t c0 = (px >> (8 * OFF)) & 0xffu;
    const uint32_t c1 = (px >> (8 * OFF + 8)) & 0xffu;
    const uint32_t c2 = (px >> (8 * OFF + 16)) & 0xffu;
    R = BGR ? c2 : c0;
    G = c1;
    B = BGR ? c0 : c2;
}

template <int OFF, bool BGR>
__device__ __forceinline__ uint32_t repack4(uint32_t px, uint32_t R, uint32_t G, uint32_t B)
{
    const uint32_t c0 = BGR ? B : R, c2 = BGR ? R : B;
    const uint32_t keep = OFF == 0 ? 0xff000000u : 0x000000ffu; // alpha / x byte untouched
    return (px & keep) | (c0 << (8 * OFF)) | (G << (8 * OFF + 8)) | (c2 << (8 * OFF + 16));
}

template <int OFF, bool BGR, int VARIANT>
__device__ __forceinline__ uint32_t filter_px4(uint32_t px, const FastConsts &k, const FilterLds &lds)
{
    if constexpr (VARIANT == kGeneral) {
        uint32_t R, G, B;
        unpack4<OFF, BGR>(px, R, G, B);
        hsvfilter_pixel<VARIANT>(R, G, B, k, lds.sextant);
        return repack4<OFF, BGR>(px, R, G, B);
    } else {
        // v_cvt_f32_ubyteN straight from the pixel dword + the exact 2-op divide.  (A 256-entry LDS
        // table of RN(b/255) was measured slower: 3 more random ds_read_b32 per pixel cost more in
        // bank conflicts than the 5 fast VALU ops they replace: 60.3 k vs 68.2 k frames/s.)
        const float c0 = div255((float)((px >> (8 * OFF)) & 0xffu), k);
        const float c1 = div255((float)((px >> (8 * OFF + 8)) & 0xffu), k);
        const float c2 = div255((float)((px >> (8 * OFF + 16)) & 0xffu), k);
        uint32_t T;
        const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(BGR ? c2 : c0, c1, BGR ? c0 : c2, k, T);
        return __builtin_amdgcn_perm(T, px, sextant_at(lds.sextant, sel_off));
    }
}

// ---- hsvfilter, 4-byte formats -------------------------------------------------------------
// width = pixels per row, rows/stride describe one frame, fb.base[blockIdx.z] its plane 0.
// NT: non-temporal loads/stores (vec4 mode), for frames that are not read again on the GPU right away.
template <int OFF, bool BGR, int VARIANT, int MODE, bool NT = false, int TILE = 1>
__global__ __launch_bounds__(kBlock) void hsvfilter4_kernel(FrameBatch fb, uint64_t width,
                                                            uint32_t rows, uint64_t stride,
                                                            FastConsts p)
{
    __shared__ FilterLds lds;
    // vec4 / dword modes permute straight into the pixel layout; the byte mode uses (off 0, RGB)
    init_filter_lds<VARIANT>(lds, MODE == kModeBytes ? 0 : OFF, MODE == kModeBytes ? false : BGR);
    const uint32_t *lut = lds.sextant;
    uint8_t *frame = fb.base[blockIdx.z];
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *line = frame + (uint64_t)row * stride;
        if constexpr (MODE == kModeVec4) {
            // One workgroup owns a tile of TILE x 256 pixel groups: every lane issues its TILE
            // 16-byte loads first, then runs the arithmetic, so the loads of group u+1.. are in
            // flight while group u is computed (the wave stays in its VALU phase for TILE x 272
            // instructions instead of dying after one group).
            const uint64_t groups = (width + 3) >> 2;
            // (giving every XCD one contiguous eighth of the frame instead of every 8th tile: -1 %)
            for (uint64_t t0 = (uint64_t)blockIdx.x * (kBlock * TILE); t0 < groups;
                 t0 += (uint64_t)gridDim.x * (kBlock * TILE)) {
                uint4 v[TILE];
#pragma unroll
                for (int u = 0; u < TILE; u++) {
                    const uint64_t x = (t0 + (uint64_t)u * kBlock + threadIdx.x) << 2;
                    if (x + 4 <= width) {
                        if constexpr (NT) {
                            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(line + x * 4));
                            v[u] = make_uint4(t.x, t.y, t.z, t.w);
                        } else {
                            v[u] = *reinterpret_cast<const uint4 *>(line + x * 4);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < TILE; u++) {
                    const uint64_t x = (t0 + (uint64_t)u * kBlock + threadIdx.x) << 2;
                    if (x + 4 <= width) {
                        v[u].x = filter_px4<OFF, BGR, VARIANT>(v[u].x, p, lds);
                        v[u].y = filter_px4<OFF, BGR, VARIANT>(v[u].y, p, lds);
                        v[u].z = filter_px4<OFF, BGR, VARIANT>(v[u].z, p, lds);
                        v[u].w = filter_px4<OFF, BGR, VARIANT>(v[u].w, p, lds);
                        if constexpr (NT) {
                            const u32x4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
                            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(line + x * 4));
                        } else {
                            *reinterpret_cast<uint4 *>(line + x * 4) = v[u];
                        }
                    } else {
                        for (uint64_t xx = x; xx < width; xx++) {
                            uint32_t *q = reinterpret_cast<uint32_t *>(line + xx * 4);
                            *q = filter_px4<OFF, BGR, VARIANT>(*q, p, lds);
                        }
                    }
                }
            }
        } else if constexpr (MODE == kModeDword) {
            for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < width;
                 x += (uint64_t)gridDim.x * kBlock) {
                uint32_t *q = reinterpret_cast<uint32_t *>(line + x * 4);
                *q = filter_px4<OFF, BGR, VARIANT>(*q, p, lds);
            }
        } else {
            for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < width;
                 x += (uint64_t)gridDim.x * kBlock) {
                uint8_t *q = line + x * 4 + OFF;
                uint32_t R = BGR ? q[2] : q[0], G = q[1], B = BGR ? q[0] : q[2];
                hsvfilter_pixel<VARIANT>(R, G, B, p, lut);
                q[0] = (uint8_t)(BGR ? B : R);
                q[1] = (uint8_t)G;
                q[2] = (uint8_t)(BGR ? R : B);
            }
        }
    }
}


// ---- hsvfilter, 4-byte formats, u8/255 by the texture unit: hsv_typed_kernels.hip (its own translation unit: LLVM's
// default scheduler suits it, the ILP strategy this file is built with suits everything here) ----
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---- hsvfilter, 3-byte formats (RGB / BGR) ---------------------------------------------------
struct __attribute__((aligned(4))) U3 {
    uint32_t a, b, c;
};

template <bool BGR, int VARIANT>
__device__ __forceinline__ void filter_triplet(uint32_t &c0, uint32_t &c1, uint32_t &c2,
                                               const FastConsts &p, const uint32_t *lut)
{
    uint32_t R = BGR ? c2 : c0, G = c1, B = BGR ? c0 : c2;
    hsvfilter_pixel<VARIANT>(R, G, B, p, lut);
    c0 = BGR ? B : R;
    c1 = G;
    c2 = BGR ? R : B;
}

template <bool BGR, int VARIANT, int MODE>
__global__ __launch_bounds__(kBlock) void hsvfilter3_kernel(FrameBatch fb, uint64_t width,
                                                            uint32_t rows, uint64_t stride,
                                                            FastConsts p)
{
    __shared__ FilterLds lds;
    init_filter_lds<VARIANT>(lds, 0, false);
    const uint32_t *lut = lds.sextant;
    uint8_t *frame = fb.base[blockIdx.z];
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *line = frame + (uint64_t)row * stride;
        if constexpr (MODE == kModeVec4) {
            const uint64_t groups = (width + 3) >> 2;
            for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups;
                 g += (uint64_t)gridDim.x * kBlock) {
                const uint64_t x = g << 2;
                if (x + 4 <= width) {
                    U3 v = *reinterpret_cast<const U3 *>(line + x * 3);
                    // 12 bytes = 4 pixels: p0 = a[0..2], p1 = a[3] b[0..1], p2 = b[2..3] c[0], p3 = c[1..3]
                    uint32_t k[12];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        k[i] = (v.a >> (8 * i)) & 0xffu;
                        k[4 + i] = (v.b >> (8 * i)) & 0xffu;
                        k[8 + i] = (v.c >> (8 * i)) & 0xffu;
                    }
#pragma unroll
                    for (int px = 0; px < 4; px++)
                        filter_triplet<BGR, VARIANT>(k[3 * px], k[3 * px + 1], k[3 * px + 2], p, lut);
                    v.a = k[0] | (k[1] << 8) | (k[2] << 16) | (k[3] << 24);
                    v.b = k[4] | (k[5] << 8) | (k[6] << 16) | (k[7] << 24);
                    v.c = k[8] | (k[9] << 8) | (k[10] << 16) | (k[11] << 24);
                    *reinterpret_cast<U3 *>(line + x * 3) = v;
                } else {
                    for (uint64_t xx = x; xx < width; xx++) {
                        uint8_t *q = line + xx * 3;
                        uint32_t c0 = q[0], c1 = q[1], c2 = q[2];
                        filter_triplet<BGR, VARIANT>(c0, c1, c2, p, lut);
                        q[0] = (uint8_t)c0; q[1] = (uint8_t)c1; q[2] = (uint8_t)c2;
                    }
                }
            }
        } else {
            for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < width;
                 x += (uint64_t)gridDim.x * kBlock) {
                uint8_t *q = line + x * 3;
                uint32_t c0 = q[0], c1 = q[1], c2 = q[2];
                filter_triplet<BGR, VARIANT>(c0, c1, c2, p, lut);
                q[0] = (uint8_t)c0; q[1] = (uint8_t)c1; q[2] = (uint8_t)c2;
            }
        }
    }
}

// ---- hsvdetector ----------------------------------------------------------------------------
// IN_BPP 3|4, IN_OFF first colour byte, IN_BGR byte order; OUT_A0 alpha first (ARGB/ABGR),
// OUT_BGR colour order of the output.  One lane per pixel group of 4 when aligned.
template <int IN_BPP, int IN_OFF, bool IN_BGR, bool OUT_A0, bool OUT_BGR, int VARIANT>
__device__ __forceinline__ uint32_t detect_px(uint32_t c0, uint32_t c1, uint32_t c2,
                                              const HsvDetectorParams &p)
{
    const uint32_t R = IN_BGR ? c2 : c0, G = c1, B = IN_BGR ? c0 : c2;
    const Hsv hsv = from_rgb<VARIANT>(R, G, B, p.consts);
    uint32_t a;
    if constexpr (VARIANT == kDetFast)
        a = ~detect_miss_mask_fast(hsv, p) & 0xffu;
    else
        a = detect_alpha_general(hsv, p);
    const uint32_t o0 = OUT_BGR ? B : R, o2 = OUT_BGR ? R : B;
    return OUT_A0 ? (a | (o0 << 8) | (G << 16) | (o2 << 24))
                  : (o0 | (G << 8) | (o2 << 16) | (a << 24));
}

// 4-byte input pixel -> output pixel in one v_perm_b32: colour bytes come straight from the source
// dword, the alpha byte from the hit mask.
template <int IN_OFF, bool IN_BGR, bool OUT_A0, bool OUT_BGR>
__device__ __forceinline__ uint32_t detect_px4_fast(uint32_t px, const HsvDetectorParams &p)
{
    const float c0 = (float)((px >> (8 * IN_OFF)) & 0xffu);
    const float c1 = (float)((px >> (8 * IN_OFF + 8)) & 0xffu);
    const float c2 = (float)((px >> (8 * IN_OFF + 16)) & 0xffu);
    const HsvN hsv = from_rgb_fast_n(IN_BGR ? c2 : c0, c1, IN_BGR ? c0 : c2, p.consts);
    const uint32_t hit = ~detect_miss_mask_fast(hsv, p); // 0xffffffff on a hit: any byte of it is the alpha
    // selector: 0..3 = bytes of px, 4 = alpha
    constexpr uint32_t iR = IN_OFF + (IN_BGR ? 2 : 0), iG = IN_OFF + 1, iB = IN_OFF + (IN_BGR ? 0 : 2);
    constexpr uint32_t o0 = OUT_BGR ? iB : iR, o2 = OUT_BGR ? iR : iB;
    constexpr uint32_t sel = OUT_A0 ? (4u | (o0 << 8) | (iG << 16) | (o2 << 24)) : (o0 | (iG << 8) | (o2 << 16) | (4u << 24));
    return __builtin_amdgcn_perm(hit, px, sel);
}


// hsvdetector with u8/255 by typed buffer loads (see hsvfilter4_typed_kernel): 4-byte inputs, strength-reduced hue
// test.  The descriptor's DST_SEL delivers (R, G, B); the v_perm selector that builds the output pixel from the raw
// input dword and the hit mask is a kernel argument (it depends on the two layouts only), so one instantiation serves
// all 16 format pairs.
#ifndef MVFX_DET_TILE
#define MVFX_DET_TILE 2 // 16-byte pixel groups per lane of hsvdetector_typed_kernel (round 5; 1 = rounds 3/4)
#endif
template <bool STREAM> // the output is not read again soon (MVFX_OPT_NONTEMPORAL): write-through stores (device_store.hpp)
__global__ __launch_bounds__(kBlock) void hsvdetector_typed_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width, uint32_t rows,
                                                                   uint64_t in_stride, uint64_t out_stride, HsvDetectorParams p,
                                                                   uint32_t word3, uint32_t frame_bytes, uint32_t perm_sel)
{
    constexpr int TILE = MVFX_DET_TILE;
    const uint64_t a = reinterpret_cast<uint64_t>(in_fb.base[blockIdx.z]);
    uint8_t *out = out_fb.base[blockIdx.z];
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)frame_bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)word3);
    const uint32_t sel = __builtin_amdgcn_readfirstlane(perm_sel);
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *oline = out + (uint64_t)row * out_stride;
        const uint32_t line_off = (uint32_t)((uint64_t)row * in_stride);
        const uint64_t groups = width >> 2; // the launcher guarantees width % 4 == 0
        for (uint64_t t0 = (uint64_t)blockIdx.x * (kBlock * TILE); t0 < groups; t0 += (uint64_t)gridDim.x * (kBlock * TILE)) {
            u32x4 raw[TILE];
            f32x3 c[TILE][4];
            uint32_t voff[TILE];
#pragma unroll
            for (int u = 0; u < TILE; u++) // groups past the end: the buffer bounds check returns zeros, nothing is stored
                voff[u] = line_off + (uint32_t)((t0 + (uint64_t)u * kBlock + threadIdx.x) << 4);
            if constexpr (TILE == 2) {
                asm volatile("buffer_load_dwordx4 %0, %10, %12, 0 offen\n\t"
                             "buffer_load_dwordx4 %1, %11, %12, 0 offen\n\t"
                             "buffer_load_format_xyz %2, %10, %12, 0 offen\n\t"
                             "buffer_load_format_xyz %3, %10, %12, 0 offen offset:4\n\t"
                             "buffer_load_format_xyz %4, %10, %12, 0 offen offset:8\n\t"
                             "buffer_load_format_xyz %5, %10, %12, 0 offen offset:12\n\t"
                             "buffer_load_format_xyz %6, %11, %12, 0 offen\n\t"
                             "buffer_load_format_xyz %7, %11, %12, 0 offen offset:4\n\t"
                             "buffer_load_format_xyz %8, %11, %12, 0 offen offset:8\n\t"
                             "buffer_load_format_xyz %9, %11, %12, 0 offen offset:12\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(raw[0]), "=&v"(raw[TILE - 1]), "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]),
                               "=&v"(c[TILE - 1][0]), "=&v"(c[TILE - 1][1]), "=&v"(c[TILE - 1][2]), "=&v"(c[TILE - 1][3])
                             : "v"(voff[0]), "v"(voff[TILE - 1]), "s"(rs)
                             : "memory");
            } else {
                asm volatile("buffer_load_dwordx4 %0, %5, %6, 0 offen\n\t"
                             "buffer_load_format_xyz %1, %5, %6, 0 offen\n\t"
                             "buffer_load_format_xyz %2, %5, %6, 0 offen offset:4\n\t"
                             "buffer_load_format_xyz %3, %5, %6, 0 offen offset:8\n\t"
                             "buffer_load_format_xyz %4, %5, %6, 0 offen offset:12\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(raw[0]), "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3])
                             : "v"(voff[0]), "s"(rs)
                             : "memory");
            }
#pragma unroll
            for (int u = 0; u < TILE; u++) {
                const uint64_t g = t0 + (uint64_t)u * kBlock + threadIdx.x;
                if (g < groups) {
                    const uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                    uint32_t r[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const HsvN hsv = from_unit_rgb_fast_n(c[u][j].x, c[u][j].y, c[u][j].z, p.consts);
                        r[j] = __builtin_amdgcn_perm(~detect_miss_mask_fast(hsv, p), w[j], sel); // selector byte 4 = the hit mask
                    }
                    stream_store16<STREAM>(oline + (g << 4), store_u32x4{r[0], r[1], r[2], r[3]});
                }
            }
        }
    }
}

// The same for 3-byte inputs (RGB / BGR -> any 4-byte output), round 5: typed loads at unaligned byte offsets (hsvfilter3_typed_kernel in
// hsv_typed_kernels.hip has the probe and the two-descriptor scheme: a lane reads its own twelve bytes only).  The output pixel's three
// colour bytes come out of the raw twelve bytes with one two-source v_perm_b32 per pixel (selectors from the host: they depend on the
// two layouts only), the alpha byte is the hit mask through one v_and_or_b32.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
template <bool STREAM>
__global__ __launch_bounds__(kBlock) void hsvdetector3_typed_kernel(FrameBatch in_fb, FrameBatch out_fb, uint64_t width, uint32_t rows,
                                                                    uint64_t in_stride, uint64_t out_stride, HsvDetectorParams p,
                                                                    uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, uint4 perm_sel,
                                                                    uint32_t alpha_mask)
{
    const uint64_t a = reinterpret_cast<uint64_t>(in_fb.base[blockIdx.z]);
    uint8_t *out = out_fb.base[blockIdx.z];
    i32x4 ra, rb;
    ra.x = rb.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    ra.y = rb.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    ra.z = rb.z = __builtin_amdgcn_readfirstlane((int)frame_bytes);
    ra.w = __builtin_amdgcn_readfirstlane((int)word3a);
    rb.w = __builtin_amdgcn_readfirstlane((int)word3b);
    const uint32_t s0 = __builtin_amdgcn_readfirstlane(perm_sel.x), s1 = __builtin_amdgcn_readfirstlane(perm_sel.y),
                   s2 = __builtin_amdgcn_readfirstlane(perm_sel.z), s3 = __builtin_amdgcn_readfirstlane(perm_sel.w);
    const uint32_t am = __builtin_amdgcn_readfirstlane(alpha_mask);
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *oline = out + (uint64_t)row * out_stride;
        const uint32_t line_off = (uint32_t)((uint64_t)row * in_stride);
        const uint64_t groups = width >> 2; // the launcher guarantees width % 4 == 0
        for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * kBlock) {
            u32x3_t raw;
            f32x3 c[4];
            const uint32_t voff = line_off + (uint32_t)g * 12u;
            asm volatile("buffer_load_dwordx3 %0, %5, %6, 0 offen\n\t"
                         "buffer_load_format_xyz %1, %5, %6, 0 offen\n\t"
                         "buffer_load_format_xyz %2, %5, %6, 0 offen offset:3\n\t"
                         "buffer_load_format_xyz %3, %5, %6, 0 offen offset:6\n\t"
                         "buffer_load_format_xyz %4, %5, %7, 0 offen offset:8\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(raw), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                         : "v"(voff), "s"(ra), "s"(rb)
                         : "memory");
            // pixel j's three bytes sit in (lo, hi) = (a, a), (a, b), (b, c), (c, c)   (v_perm: selector bytes 0-3 = S1 = lo, 4-7 = S0 = hi)
            const uint32_t lo[4] = {raw.x, raw.x, raw.y, raw.z}, hi[4] = {raw.x, raw.y, raw.z, raw.z}, sel[4] = {s0, s1, s2, s3};
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const HsvN hsv = from_unit_rgb_fast_n(c[j].x, c[j].y, c[j].z, p.consts);
                const uint32_t colour = __builtin_amdgcn_perm(hi[j], lo[j], sel[j]);
                r[j] = (~detect_miss_mask_fast(hsv, p) & am) | colour;
            }
            stream_store16<STREAM>(oline + (g << 4), store_u32x4{r[0], r[1], r[2], r[3]});
        }
    }
}

template <int IN_BPP, int IN_OFF, bool IN_BGR, bool OUT_A0, bool OUT_BGR, int VARIANT, int MODE>
__global__ __launch_bounds__(kBlock) void hsvdetector_kernel(FrameBatch in_fb, FrameBatch out_fb,
                                                             uint64_t width, uint32_t rows,
                                                             uint64_t in_stride,
                                                             uint64_t out_stride,
                                                             HsvDetectorParams p)
{
    const uint8_t *in = in_fb.base[blockIdx.z]; // one frame pair of the batch per grid z
    uint8_t *out = out_fb.base[blockIdx.z];
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const uint8_t *iline = in + (uint64_t)row * in_stride;
        uint8_t *oline = out + (uint64_t)row * out_stride;
        if constexpr (MODE == kModeVec4) {
            const uint64_t groups = (width + 3) >> 2;
            for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups;
                 g += (uint64_t)gridDim.x * kBlock) {
                const uint64_t x = g << 2;
                if (x + 4 <= width) {
                    uint4 o;
                    if constexpr (IN_BPP == 4) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(iline + x * 4);
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                        uint32_t r[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            if constexpr (VARIANT == kDetFast)
                                r[i] = detect_px4_fast<IN_OFF, IN_BGR, OUT_A0, OUT_BGR>(w[i], p);
                            else
                                r[i] = detect_px<IN_BPP, IN_OFF, IN_BGR, OUT_A0, OUT_BGR, VARIANT>(
                                    (w[i] >> (8 * IN_OFF)) & 0xffu, (w[i] >> (8 * IN_OFF + 8)) & 0xffu,
                                    (w[i] >> (8 * IN_OFF + 16)) & 0xffu, p);
                        }
                        o = make_uint4(r[0], r[1], r[2], r[3]);
                    } else {
                        const U3 v = *reinterpret_cast<const U3 *>(iline + x * 3);
                        uint32_t k[12];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            k[i] = (v.a >> (8 * i)) & 0xffu;
                            k[4 + i] = (v.b >> (8 * i)) & 0xffu;
                            k[8 + i] = (v.c >> (8 * i)) & 0xffu;
                        }
                        uint32_t r[4];
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            r[i] = detect_px<IN_BPP, IN_OFF, IN_BGR, OUT_A0, OUT_BGR, VARIANT>(
                                k[3 * i], k[3 * i + 1], k[3 * i + 2], p);
                        o = make_uint4(r[0], r[1], r[2], r[3]);
                    }
                    *reinterpret_cast<uint4 *>(oline + x * 4) = o;
                } else {
                    for (uint64_t xx = x; xx < width; xx++) {
                        const uint8_t *q = iline + xx * IN_BPP + IN_OFF;
                        const uint32_t r = detect_px<IN_BPP, IN_OFF, IN_BGR, OUT_A0, OUT_BGR, VARIANT>(
                            q[0], q[1], q[2], p);
                        *reinterpret_cast<uint32_t *>(oline + xx * 4) = r;
                    }
                }
            }
        } else {
            for (uint64_t x = (uint64_t)blockIdx.x * kBlock + threadIdx.x; x < width;
                 x += (uint64_t)gridDim.x * kBlock) {
                const uint8_t *q = iline + x * IN_BPP + IN_OFF;
                const uint32_t r = detect_px<IN_BPP, IN_OFF, IN_BGR, OUT_A0, OUT_BGR, VARIANT>(
                    q[0], q[1], q[2], p);
                uint8_t *o = oline + x * 4;
                o[0] = (uint8_t)r; o[1] = (uint8_t)(r >> 8); o[2] = (uint8_t)(r >> 16); o[3] = (uint8_t)(r >> 24);
            }
        }
    }
}

// ---- f32 HSV dump (tests) -------------------------------------------------------------------
template <int OFF, bool BGR, int VARIANT>
__global__ __launch_bounds__(kBlock) void hsv_from_frame_kernel(const uint8_t *in, float *out,
                                                                uint32_t width, uint32_t rows,
                                                                uint64_t stride, FastConsts k)
{
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        for (uint32_t x = blockIdx.x * kBlock + threadIdx.x; x < width; x += gridDim.x * kBlock) {
            const uint8_t *q = in + (uint64_t)row * stride + (uint64_t)x * 4 + OFF;
            const uint32_t R = BGR ? q[2] : q[0], G = q[1], B = BGR ? q[0] : q[2];
            const Hsv hsv = from_rgb<VARIANT>(R, G, B, k);
            float *o = out + ((uint64_t)row * width + x) * 3;
            o[0] = hsv.h; o[1] = hsv.s; o[2] = hsv.v;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------

// Kernel choices come from the calling thread's options (mvfx_thread_set_options, capi_common.hip): no process globals,
// so two elements on two streaming threads never see each other's choice.

// Domain of the FAST kernels (hsv_math.hpp): finite settings, |shift| <= 360, shift not in
// (0,1e-30) in magnitude.
bool fast_domain_ok(const mvfx_hsvfilter_settings &s)
{
    const float v[5] = {s.hue_shift, s.saturation_mul, s.saturation_off, s.value_mul, s.value_off};
    for (float f : v)
        if (!std::isfinite(f))
            return false;
    const float a = std::fabs(s.hue_shift);
    return a <= 360.0f && (a == 0.0f || a >= 1e-30f);
}

FastConsts make_consts(const mvfx_hsvfilter_settings *s)
{
    FastConsts k{};
    k.c255 = 1.0f / 255.0f;
    k.c255lo = (float)(1.0 / 255.0 - (double)k.c255);
    k.c60 = 1.0f / 60.0f;
    k.c60lo = (float)(1.0 / 60.0 - (double)k.c60);
    k.c120 = 0.5f * k.c60;
    k.c120lo = 0.5f * k.c60lo;
    k.sext_magic = 1048575.9375f;
    k.k255 = 255.0f;
    k.k60 = 60.0f;
    k.k360 = 360.0f;
    k.neg_k60 = -60.0f;
    k.pred360 = std::nextafterf(360.0f, 0.0f);
    k.tiny = 1e-30f;
    const float f360 = 360.0f;
    std::memcpy(&k.bits360, &f360, 4);
    if (s) {
        k.hue_shift = s->hue_shift;
        k.saturation_mul = s->saturation_mul;
        k.saturation_off = s->saturation_off;
        k.value_mul = s->value_mul;
        k.value_off = s->value_off;
        k.neg_saturation_mul = -s->saturation_mul;
    }
    return k;
}

int filter_layout(int format, int *bpp, int *off, bool *bgr)
{
    switch (format) {
    case MVFX_FORMAT_RGBX: case MVFX_FORMAT_RGBA: *bpp = 4; *off = 0; *bgr = false; return 0;
    case MVFX_FORMAT_XRGB: case MVFX_FORMAT_ARGB: *bpp = 4; *off = 1; *bgr = false; return 0;
    case MVFX_FORMAT_BGRX: case MVFX_FORMAT_BGRA: *bpp = 4; *off = 0; *bgr = true; return 0;
    case MVFX_FORMAT_XBGR: case MVFX_FORMAT_ABGR: *bpp = 4; *off = 1; *bgr = true; return 0;
    case MVFX_FORMAT_RGB: *bpp = 3; *off = 0; *bgr = false; return 0;
    case MVFX_FORMAT_BGR: *bpp = 3; *off = 0; *bgr = true; return 0;
    default: return -1;
    }
}

struct Geometry {
    uint64_t width;  // pixels per logical row
    uint32_t rows;
    uint64_t stride;
    int mode;
    int tile; // pixel groups per lane (vec4 mode)
    dim3 grid;
};

// Picks flat/row layout, access mode and grid for a packed frame batch.
Geometry plan(const mvfx_frame *frames, uint32_t n, int bpp, uint32_t n_frames_z, int tile)
{
    Geometry g;
    const mvfx_frame &f = frames[0];
    uint64_t base_or = 0;
    for (uint32_t i = 0; i < n; i++)
        base_or |= (uint64_t)(uintptr_t)frames[i].data;
    const uint64_t align_or = base_or | f.stride;
    const uint64_t need = bpp == 4 ? 15 : 3;
    const bool flat = (uint64_t)f.width * bpp == f.stride;
    if (flat) {
        g.width = (uint64_t)f.width * f.height;
        g.rows = 1;
        g.stride = 0;
        g.mode = (base_or & need) == 0 ? kModeVec4 : ((bpp == 4 && (base_or & 3) == 0) ? kModeDword : kModeBytes);
    } else {
        g.width = f.width;
        g.rows = f.height;
        g.stride = f.stride;
        g.mode = (align_or & need) == 0 ? kModeVec4 : ((bpp == 4 && (align_or & 3) == 0) ? kModeDword : kModeBytes);
    }
    const uint64_t work = g.mode == kModeVec4 ? (g.width + 3) / 4 : g.width;
    g.tile = 1;
    if (g.mode == kModeVec4 && tile > 1 && work * n_frames_z >= kTileMinGroups)
        g.tile = tile;
    const uint64_t per_block = (uint64_t)kBlock * g.tile;
    uint64_t bx = (work + per_block - 1) / per_block;
    if (bx == 0) bx = 1;
    if (bx > 65535u * 16u) bx = 65535u * 16u; // grid-stride covers the rest
    g.grid = dim3((uint32_t)bx, g.rows < 65535u ? (g.rows ? g.rows : 1) : 65535u, n_frames_z);
    return g;
}

template <int VARIANT>
void launch_filter(int bpp, int off, bool bgr, const Geometry &g, const FrameBatch &fb,
                   const FastConsts &p, hipStream_t stream)
{
    const bool nontemporal = opt_nontemporal();

#define MVFX_L4(O, B, M) \
    MVFX_LAUNCH((hsvfilter4_kernel<O, B, VARIANT, M>), g.grid, dim3(kBlock), 0, stream, fb, g.width, g.rows, g.stride, p)
#define MVFX_L4V(O, B, NT_) \
    do { if (g.tile == kTile) MVFX_LAUNCH((hsvfilter4_kernel<O, B, VARIANT, kModeVec4, NT_, kTile>), g.grid, dim3(kBlock), 0, stream, fb, g.width, g.rows, g.stride, p); \
         else MVFX_LAUNCH((hsvfilter4_kernel<O, B, VARIANT, kModeVec4, NT_, 1>), g.grid, dim3(kBlock), 0, stream, fb, g.width, g.rows, g.stride, p); } while (0)
#define MVFX_L3(B, M) \
    MVFX_LAUNCH((hsvfilter3_kernel<B, VARIANT, M>), g.grid, dim3(kBlock), 0, stream, fb, g.width, g.rows, g.stride, p)
    if (bpp == 4) {
        const int key = (off ? 2 : 0) | (bgr ? 1 : 0);
        switch (g.mode) {
        case kModeVec4:
            if (nontemporal && VARIANT != kGeneral)
                switch (key) { case 0: MVFX_L4V(0, false, true); break; case 1: MVFX_L4V(0, true, true); break;
                               case 2: MVFX_L4V(1, false, true); break; default: MVFX_L4V(1, true, true); break; }
            else
                switch (key) { case 0: MVFX_L4V(0, false, false); break; case 1: MVFX_L4V(0, true, false); break;
                               case 2: MVFX_L4V(1, false, false); break; default: MVFX_L4V(1, true, false); break; }
            break;
        case kModeDword:
            switch (key) { case 0: MVFX_L4(0, false, kModeDword); break; case 1: MVFX_L4(0, true, kModeDword); break;
                           case 2: MVFX_L4(1, false, kModeDword); break; default: MVFX_L4(1, true, kModeDword); break; }
            break;
        default:
            switch (key) { case 0: MVFX_L4(0, false, kModeBytes); break; case 1: MVFX_L4(0, true, kModeBytes); break;
                           case 2: MVFX_L4(1, false, kModeBytes); break; default: MVFX_L4(1, true, kModeBytes); break; }
            break;
        }
    } else {
        if (g.mode == kModeVec4) { if (bgr) MVFX_L3(true, kModeVec4); else MVFX_L3(false, kModeVec4); }
        else { if (bgr) MVFX_L3(true, kModeBytes); else MVFX_L3(false, kModeBytes); }
    }
#undef MVFX_L4
#undef MVFX_L4V
#undef MVFX_L3
}


// ---- hsvfilter on an I420 frame: `videoconvert ! hsvfilter ! videoconvert` in one kernel -------------------------
// Decoders hand over I420; the reference's hsvfilter takes RGB only (hsvfilter/imp.rs:278-289), so every real pipeline
// wraps it in two videoconverts.  Fused (tile walk: convert_math.hpp i420_fused_tile): 1.5 B/px read + 1.5 B/px written
// instead of 5.5 + 8 + 5.5 through three launches; the RGBA pixel exists only in a register.
template <int VARIANT>
__global__ __launch_bounds__(kI420Block) void hsvfilter_i420_kernel(I420Planes pl, uint32_t width, uint32_t height, FastConsts p,
                                                                    YuvToRgbCoef kin, RgbToYuvCoef kout)
{
    __shared__ FilterLds lds;
    __shared__ int2 edge[kI420Block];
    init_filter_lds<VARIANT>(lds, 0, false); // RGBA register layout: colour bytes 0..2, RGB order
    i420_fused_tile(pl, width, height, kin, kout, edge, [&](uint32_t px) { return filter_px4<0, false, VARIANT>(px, p, lds); });
}

int hsvfilter_impl(const mvfx_frame *frames, uint32_t n, const mvfx_hsvfilter_settings *s,
                   hipStream_t stream)
{
    if (!frames || !s || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: NULL frame/settings or empty batch");
    int bpp, off;
    bool bgr;
    if (filter_layout(frames[0].format, &bpp, &off, &bgr) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT,
                    "hsvfilter: format %d is not one of RGBx xRGB BGRx xBGR RGBA ARGB BGRA ABGR RGB BGR "
                    "(hsvfilter/imp.rs:372 unreachable!())", frames[0].format);
    for (uint32_t i = 0; i < n; i++) {
        int rc = check_packed_frame(&frames[i], "hsvfilter");
        if (rc != MVFX_OK)
            return rc;
        if (frames[i].width != frames[0].width || frames[i].height != frames[0].height ||
            frames[i].stride != frames[0].stride || frames[i].format != frames[0].format)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: frames of one batch must share geometry and format");
    }
    // assert_eq!(data.len() % nb_channels, 0) hsvfilter/imp.rs:92
    if (((uint64_t)frames[0].stride * frames[0].height) % (uint64_t)bpp != 0)
        return fail(MVFX_ERR_REFERENCE_PANIC,
                    "hsvfilter: plane size %llu is not a multiple of %d bytes per pixel; the reference "
                    "asserts on this (hsvfilter/imp.rs:92)",
                    (unsigned long long)((uint64_t)frames[0].stride * frames[0].height), bpp);
    if (int rc = require_device(); rc != MVFX_OK)
        return rc;

    const bool fast_ok = fast_domain_ok(*s);
    const int g_variant = opt_hsv_variant();
    if (g_variant == 2 && !fast_ok)
        return fail(MVFX_ERR_INVALID_ARGUMENT,
                    "hsvfilter: settings are outside the proven domain of the strength-reduced kernel");
    const bool use_fast = g_variant == 2 || (g_variant == 0 && fast_ok);
    const FastConsts p = make_consts(s);

    if (frames[0].width == 0 || frames[0].height == 0)
        return MVFX_OK;
    for (uint32_t done = 0; done < n; done += kMaxBatch) {
        const uint32_t m = (n - done) < (uint32_t)kMaxBatch ? (n - done) : (uint32_t)kMaxBatch;
        FrameBatch fb{};
        for (uint32_t i = 0; i < m; i++)
            fb.base[i] = static_cast<uint8_t *>(frames[done + i].data);
        const Geometry g = plan(frames + done, m, bpp, m, bpp == 4 ? kTile : 1);
        const uint64_t frame_bytes = (uint64_t)frames[0].stride * frames[0].height;
        const bool typed4 = opt_typed_loads() && use_fast && bpp == 4 && g.mode == kModeVec4 && (g.width & 3) == 0 && frame_bytes < (1ull << 32);
        if (opt_direct_only() && !(typed4 && m == 1 && n == 1 && g.rows == 1 && completion_event()))
            return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "hsvfilter: not a frame the direct-dispatch lane takes (one unpadded 4-byte frame, strength-reduced settings, a completion event)");
        if (typed4) {
            // descriptor word 3: DST_SEL x/y/z = the bytes holding R, G, B (4 + byte index), w = 0; NUM_FORMAT UNORM (0);
            // DATA_FORMAT 8_8_8_8 (10)
            const uint32_t iR = off + (bgr ? 2 : 0), iG = off + 1, iB = off + (bgr ? 0 : 2);
            const uint32_t word3 = (4 + iR) | ((4 + iG) << 3) | ((4 + iB) << 6) | (10u << 15);
            const bool neg = std::signbit(s->hue_shift) && s->hue_shift != 0.0f;
            // the direct-dispatch lane (direct_dispatch.h): ONE flat frame whose dependencies have finished and whose completion the caller takes
            // from the thread's completion event goes out as an AQL packet without the barrier bit on the library's own queue
            if (m == 1 && n == 1 && opt_direct() && g.rows == 1 && completion_event()) {
                DirectHsvArgs da{};
                da.frame = fb.base[0];
                da.groups = (uint32_t)(g.width / 4);
                da.word3 = word3;
                da.frame_bytes = (uint32_t)frame_bytes;
                da.off = off;
                da.bgr = bgr ? 1 : 0;
                da.p = p;
                const int drc = direct_hsvfilter_submit(da, neg, opt_nontemporal(), direct_queue_hint(stream));
                if (drc == MVFX_OK) continue;
                if (drc < 0) return drc;
                // (1: the lane is not available here -- the launch below, on `stream`)
            }
            if (opt_direct_only()) return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "hsvfilter: the direct-dispatch lane cannot take this frame");
            launch_hsvfilter_typed(neg, g.tile == kTile ? kTile : 1, opt_nontemporal(), g.grid, stream, fb, g.width, g.rows, g.stride, p, word3,
                                   (uint32_t)frame_bytes, off, bgr);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        if (opt_typed_loads() && use_fast && bpp == 3 && g.mode == kModeVec4 && (g.width & 3) != 0 && g.rows > 1 && frame_bytes < (1ull << 32) &&
            (uint64_t)g.rows * (g.width / 4) * (g.width / 4) < (1ull << 32)) {
            // a width that is not a multiple of four (the frame is row-padded: plan() did not flatten it): the frame's groups as ONE index space +
            // a lane per trailing pixel (hsvfilter3_typed_rows_kernel; the condition above keeps its reciprocal division exact)
            const uint32_t r0 = bgr ? 2 : 0, b0 = bgr ? 0 : 2;
            const uint32_t word3a = (4 + r0) | (5u << 3) | ((4 + b0) << 6) | (10u << 15), word3b = (5 + r0) | (6u << 3) | ((5 + b0) << 6) | (10u << 15);
            const bool neg = std::signbit(s->hue_shift) && s->hue_shift != 0.0f;
            launch_hsvfilter3_typed_rows(neg, opt_nontemporal(), m, stream, fb, (uint32_t)g.width, g.rows, (uint32_t)g.stride, p, word3a, word3b, (uint32_t)frame_bytes, bgr);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        if (opt_typed_loads() && use_fast && bpp == 3 && g.mode == kModeVec4 && (g.width & 3) == 0 && frame_bytes < (1ull << 32)) {
            // RGB / BGR by typed loads at unaligned byte offsets (hsvfilter3_typed_kernel): descriptor A delivers bytes 0, 1, 2 of the four
            // fetched, descriptor B bytes 1, 2, 3 (a lane's fourth pixel, read from byte offset 8 of its twelve)
            const uint32_t r0 = bgr ? 2 : 0, b0 = bgr ? 0 : 2;
            const uint32_t word3a = (4 + r0) | (5u << 3) | ((4 + b0) << 6) | (10u << 15), word3b = (5 + r0) | (6u << 3) | ((5 + b0) << 6) | (10u << 15);
            const bool neg = std::signbit(s->hue_shift) && s->hue_shift != 0.0f;
            const int tile3 = (g.width / 4) * m >= kTileMinGroups ? kTile : 1;
            dim3 grid3 = g.grid;
            if (tile3 > 1) grid3.x = (uint32_t)std::max<uint64_t>(1, (g.width / 4 + (uint64_t)kBlock * tile3 - 1) / ((uint64_t)kBlock * tile3));
            launch_hsvfilter3_typed(neg, tile3, opt_nontemporal(), grid3, stream, fb, g.width, g.rows, g.stride, p, word3a, word3b, (uint32_t)frame_bytes, bgr);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        if (use_fast && std::signbit(s->hue_shift) && s->hue_shift != 0.0f)
            launch_filter<kFastNeg>(bpp, off, bgr, g, fb, p, stream);
        else if (use_fast)
            launch_filter<kFast>(bpp, off, bgr, g, fb, p, stream);
        else
            launch_filter<kGeneral>(bpp, off, bgr, g, fb, p, stream);
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}


// n frames of one geometry and format, every frame with ITS OWN settings (frames of different hsvfilter elements that the launch
// combiner put together).  Frames whose settings allow the typed strength-reduced kernel share launches (one per sign of
// hue-shift: the wrap of the shifted hue is compiled in); the others go through hsvfilter_impl one by one.  Same bytes as n calls of
// hsvfilter_impl.
int hsvfilter_frames_impl(const mvfx_frame *frames, uint32_t n, const mvfx_hsvfilter_settings *settings, hipStream_t stream)
{
    if (!frames || !settings || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: NULL frame/settings or empty batch");
    int bpp, off;
    bool bgr;
    if (filter_layout(frames[0].format, &bpp, &off, &bgr) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvfilter: format %d is not a packed RGB format (hsvfilter/imp.rs:372)", frames[0].format);
    bool same = true;
    for (uint32_t i = 1; i < n; i++)
        same = same && frames[i].width == frames[0].width && frames[i].height == frames[0].height && frames[i].stride == frames[0].stride &&
               frames[i].format == frames[0].format;
    const uint64_t frame_bytes = (uint64_t)frames[0].stride * frames[0].height;
    const bool typed_ok = same && opt_typed_loads() && opt_hsv_variant() != 1 && bpp == 4 && frame_bytes < (1ull << 32) && frame_bytes % 4 == 0 &&
                          frames[0].width != 0 && frames[0].height != 0;
    std::vector<uint32_t> group[2]; // [hue_shift negative]
    for (uint32_t i = 0; i < n; i++) {
        const mvfx_hsvfilter_settings &s = settings[i];
        if (typed_ok && fast_domain_ok(s) && check_packed_frame(&frames[i], "hsvfilter") == MVFX_OK)
            group[std::signbit(s.hue_shift) && s.hue_shift != 0.0f ? 1 : 0].push_back(i);
        else if (int rc = hsvfilter_impl(&frames[i], 1, &s, stream); rc != MVFX_OK)
            return rc;
    }
    if (group[0].empty() && group[1].empty()) return MVFX_OK;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t iR = off + (bgr ? 2 : 0), iG = off + 1, iB = off + (bgr ? 0 : 2);
    const uint32_t word3 = (4 + iR) | ((4 + iG) << 3) | ((4 + iB) << 6) | (10u << 15);
    for (int neg = 0; neg < 2; neg++) {
        const std::vector<uint32_t> &idx = group[neg];
        for (size_t done = 0; done < idx.size(); done += kMaxBatch) {
            const uint32_t m = (uint32_t)std::min<size_t>(idx.size() - done, kMaxBatch);
            FrameBatch fb{};
            FrameSettingsBatch fs{};
            std::vector<mvfx_frame> sel(m);
            for (uint32_t k = 0; k < m; k++) {
                const uint32_t i = idx[done + k];
                sel[k] = frames[i];
                fb.base[k] = static_cast<uint8_t *>(frames[i].data);
                const FastConsts c = make_consts(&settings[i]);
                fs.s[k] = {c.hue_shift, c.saturation_mul, c.saturation_off, c.value_mul, c.value_off, c.neg_saturation_mul};
            }
            const Geometry g = plan(sel.data(), m, bpp, m, kTile);
            if (g.mode != kModeVec4 || (g.width & 3) != 0) { // unaligned frames: the per-frame path
                for (uint32_t k = 0; k < m; k++)
                    if (int rc = hsvfilter_impl(&sel[k], 1, &settings[idx[done + k]], stream); rc != MVFX_OK) return rc;
                continue;
            }
            const FastConsts p = make_consts(&settings[idx[done]]);
            launch_hsvfilter_typed_frames(neg != 0, g.tile == kTile ? kTile : 1, opt_nontemporal(), g.grid, stream, fb, g.width, g.rows, g.stride, p, fs,
                                          word3, (uint32_t)frame_bytes, off, bgr);
            MVFX_HIP_TRY(hipGetLastError());
        }
    }
    return MVFX_OK;
}


int hsvfilter_i420_impl(const mvfx_planar_frame *in, const mvfx_planar_frame *out, const mvfx_hsvfilter_settings *s, int yuv_standard,
                        hipStream_t stream)
{
    if (!in || !out || !s)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter_i420: NULL frame or settings");
    if (in->format != MVFX_FORMAT_I420 || out->format != MVFX_FORMAT_I420)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvfilter_i420: both frames must be I420");
    if (in->width != out->width || in->height != out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "hsvfilter_i420: input %ux%u and output %ux%u differ", in->width, in->height, out->width, out->height);
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter_i420: yuv_standard %d is not 0..3", yuv_standard);
    const uint32_t w = in->width, h = in->height;
    if ((w & 1) || (h & 1))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter_i420: odd-sized frame %ux%u (RGBA -> I420 needs even sizes)", w, h);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (w == 0 || h == 0) return MVFX_OK;
    uint64_t bits = 0;
    for (int pidx = 0; pidx < 3; pidx++) {
        const uint32_t need = pidx == 0 ? w : w / 2;
        if (!in->data[pidx] || !out->data[pidx] || in->stride[pidx] < need || out->stride[pidx] < need)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter_i420: bad plane %d", pidx);
        if (in->data[pidx] == out->data[pidx])
            return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter_i420: input and output planes must not alias (the co-sited chroma filter reads neighbour input pixels)");
        const uint64_t v = (uint64_t)(uintptr_t)in->data[pidx] | in->stride[pidx] | (uint64_t)(uintptr_t)out->data[pidx] | out->stride[pidx];
        bits |= pidx == 0 ? (v & 7) : (v & 3);
    }
    const bool fast_ok = fast_domain_ok(*s);
    const int g_variant = opt_hsv_variant();
    if (g_variant == 2 && !fast_ok)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: settings are outside the proven domain of the strength-reduced kernel");
    const bool use_fast = g_variant == 2 || (g_variant == 0 && fast_ok);
    if (bits == 0 && (w % 8) == 0 && h / 2 <= 65535u) {
        const FastConsts p = make_consts(s);
        const I420Planes pl{static_cast<const uint8_t *>(in->data[0]), static_cast<const uint8_t *>(in->data[1]), static_cast<const uint8_t *>(in->data[2]),
                            static_cast<uint8_t *>(out->data[0]), static_cast<uint8_t *>(out->data[1]), static_cast<uint8_t *>(out->data[2]),
                            in->stride[0], in->stride[1], in->stride[2], out->stride[0], out->stride[1], out->stride[2]};
        const int std_ = pick_yuv_standard(h, yuv_standard);
        const YuvToRgbCoef kin = yuv_to_rgb_coef(std_);
        const RgbToYuvCoef kout = rgb_to_yuv_coef(std_);
        const dim3 grid((w / 8 + kI420Block - 1) / kI420Block, h / 2);
        if (use_fast && std::signbit(s->hue_shift) && s->hue_shift != 0.0f)
            MVFX_LAUNCH(hsvfilter_i420_kernel<kFastNeg>, grid, dim3(kI420Block), 0, stream, pl, w, h, p, kin, kout);
        else if (use_fast)
            MVFX_LAUNCH(hsvfilter_i420_kernel<kFast>, grid, dim3(kI420Block), 0, stream, pl, w, h, p, kin, kout);
        else
            MVFX_LAUNCH(hsvfilter_i420_kernel<kGeneral>, grid, dim3(kI420Block), 0, stream, pl, w, h, p, kin, kout);
        MVFX_HIP_TRY(hipGetLastError());
        return MVFX_OK;
    }
    // frames the tile walk does not cover: the same three steps through one RGBA scratch frame
    void *ra = nullptr;
    if (int rc = host_scratch((size_t)w * 4 * h, 0, &ra); rc != MVFX_OK) return rc;
    mvfx_frame fa{ra, w, h, w * 4, MVFX_FORMAT_RGBA};
    if (int rc = mvfx_convert_i420_to_rgba(in, &fa, yuv_standard, reinterpret_cast<mvfx_stream>(stream)); rc != MVFX_OK) return rc;
    if (int rc = hsvfilter_impl(&fa, 1, s, stream); rc != MVFX_OK) return rc;
    return mvfx_convert_rgba_to_i420(&fa, out, yuv_standard, reinterpret_cast<mvfx_stream>(stream));
}

int detect_in_layout(int format, int *bpp, int *off, bool *bgr)
{
    switch (format) { // hsvdetector/imp.rs:78-87
    case MVFX_FORMAT_RGBX: *bpp = 4; *off = 0; *bgr = false; return 0;
    case MVFX_FORMAT_XRGB: *bpp = 4; *off = 1; *bgr = false; return 0;
    case MVFX_FORMAT_BGRX: *bpp = 4; *off = 0; *bgr = true; return 0;
    case MVFX_FORMAT_XBGR: *bpp = 4; *off = 1; *bgr = true; return 0;
    case MVFX_FORMAT_RGB: *bpp = 3; *off = 0; *bgr = false; return 0;
    case MVFX_FORMAT_BGR: *bpp = 3; *off = 0; *bgr = true; return 0;
    default: return -1;
    }
}

int detect_out_layout(int format, bool *a0, bool *bgr)
{
    switch (format) { // hsvdetector/imp.rs:89-96
    case MVFX_FORMAT_RGBA: *a0 = false; *bgr = false; return 0;
    case MVFX_FORMAT_ARGB: *a0 = true; *bgr = false; return 0;
    case MVFX_FORMAT_BGRA: *a0 = false; *bgr = true; return 0;
    case MVFX_FORMAT_ABGR: *a0 = true; *bgr = true; return 0;
    default: return -1;
    }
}

template <int IN_BPP, int IN_OFF, bool IN_BGR, int VARIANT, int MODE>
void launch_detect_out(bool a0, bool obgr, dim3 grid, hipStream_t stream, const FrameBatch &in,
                       const FrameBatch &out, uint64_t width, uint32_t rows, uint64_t is, uint64_t os,
                       const HsvDetectorParams &p)
{
#define MVFX_LD(A, B) \
    MVFX_LAUNCH((hsvdetector_kernel<IN_BPP, IN_OFF, IN_BGR, A, B, VARIANT, MODE>), grid, dim3(kBlock), 0, stream, in, out, width, rows, is, os, p)
    if (a0) { if (obgr) MVFX_LD(true, true); else MVFX_LD(true, false); }
    else { if (obgr) MVFX_LD(false, true); else MVFX_LD(false, false); }
#undef MVFX_LD
}

template <int VARIANT, int MODE>
void launch_detect(int bpp, int off, bool ibgr, bool a0, bool obgr, dim3 grid, hipStream_t stream,
                   const FrameBatch &in, const FrameBatch &out, uint64_t width, uint32_t rows, uint64_t is,
                   uint64_t os, const HsvDetectorParams &p)
{
#define MVFX_ARGS a0, obgr, grid, stream, in, out, width, rows, is, os, p
    if (bpp == 4) {
        if (off == 0) { if (ibgr) launch_detect_out<4, 0, true, VARIANT, MODE>(MVFX_ARGS); else launch_detect_out<4, 0, false, VARIANT, MODE>(MVFX_ARGS); }
        else { if (ibgr) launch_detect_out<4, 1, true, VARIANT, MODE>(MVFX_ARGS); else launch_detect_out<4, 1, false, VARIANT, MODE>(MVFX_ARGS); }
    } else {
        if (ibgr) launch_detect_out<3, 0, true, VARIANT, MODE>(MVFX_ARGS); else launch_detect_out<3, 0, false, VARIANT, MODE>(MVFX_ARGS);
    }
#undef MVFX_ARGS
}


// ---- hsvdetector on an I420 frame: `videoconvert ! hsvdetector` in one kernel ------------------------------------
// I420 (decoder output) -> RGB in registers (convert_math.hpp) -> hsv_detect -> the detector's 4-byte output format:
// 1.5 B/px read + 4 B/px written instead of 5.5 + 8.  One lane = COLS x 2 pixels; COLS = 4: one contiguous 16-byte store per row and
// lane, as in the I420 -> RGBA converter.
template <bool OUT_A0, bool OUT_BGR, int VARIANT, bool ALIGNED, int COLS>
__global__ __launch_bounds__(kBlock) void hsvdetector_i420_kernel(const uint8_t *yp, const uint8_t *up, const uint8_t *vp, uint64_t ys,
                                                                  uint64_t us, uint64_t vs, uint32_t width, uint32_t height,
                                                                  YuvToRgbCoef k, HsvDetectorParams p, uint8_t *out, uint64_t out_stride)
{
    const uint32_t x0 = (blockIdx.x * kBlock + threadIdx.x) * COLS;
    const uint32_t y0 = blockIdx.y * 2;
    if (x0 >= width) return;
    const bool row1 = y0 + 1 < height;
    const uint8_t *yr0 = yp + (uint64_t)y0 * ys, *yr1 = yr0 + ys;
    const uint8_t *ur = up + (uint64_t)blockIdx.y * us, *vr = vp + (uint64_t)blockIdx.y * vs;
    uint8_t *o0 = out + (uint64_t)y0 * out_stride, *o1 = o0 + out_stride;
    auto detect = [&](uint32_t rgba) -> uint32_t { // rgba: R,G,B in bytes 0..2 (what videoconvert hands to the detector as RGBx)
        if constexpr (VARIANT == kDetFast)
            return detect_px4_fast<0, false, OUT_A0, OUT_BGR>(rgba, p);
        else
            return detect_px<4, 0, false, OUT_A0, OUT_BGR, VARIANT>(rgba & 0xffu, (rgba >> 8) & 0xffu, (rgba >> 16) & 0xffu, p);
    };
    // (the literal variant -- fmodf loops -- keeps the per-pixel path: sixteen inlined copies of it crash the register
    // allocator of this compiler, and it only runs for settings outside the strength-reduced domain)
    if constexpr (ALIGNED && VARIANT != kGeneral) if (x0 + COLS <= width) {
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
            const int sh = (2 * j & 3) * 8;
            pa[2 * j] = detect(yuv_pixel((wa >> sh) & 0xffu, c, k));
            pa[2 * j + 1] = detect(yuv_pixel((wa >> (sh + 8)) & 0xffu, c, k));
            pb[2 * j] = detect(yuv_pixel((wb >> sh) & 0xffu, c, k));
            pb[2 * j + 1] = detect(yuv_pixel((wb >> (sh + 8)) & 0xffu, c, k));
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
#pragma unroll 1
    for (uint32_t x = x0; x < min(x0 + COLS, width); x++) {
        const ChromaTerms c = chroma_terms(ur[x / 2], vr[x / 2], k);
#pragma unroll 1
        for (int r = 0; r < (row1 ? 2 : 1); r++) {
            const uint32_t q = detect(yuv_pixel((r ? yr1 : yr0)[x], c, k));
            uint8_t *d = (r ? o1 : o0) + (uint64_t)x * 4;
            d[0] = (uint8_t)q; d[1] = (uint8_t)(q >> 8); d[2] = (uint8_t)(q >> 16); d[3] = (uint8_t)(q >> 24);
        }
    }
}

// One frame pair of a batch: formats, sizes, the reference's asserts.
int check_detect_pair(const mvfx_frame *in, const mvfx_frame *out, int *bpp, int *off, bool *ibgr, bool *a0, bool *obgr)
{
    if (detect_in_layout(in->format, bpp, off, ibgr) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvdetector: input format %d not in RGBx xRGB BGRx xBGR RGB BGR (hsvdetector/imp.rs:78-87)", in->format);
    if (detect_out_layout(out->format, a0, obgr) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvdetector: output format %d not in RGBA ARGB BGRA ABGR (hsvdetector/imp.rs:89-96)", out->format);
    if (int rc = check_packed_frame(in, "hsvdetector input"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(out, "hsvdetector output"); rc != MVFX_OK) return rc;
    if (in->width != out->width || in->height != out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "hsvdetector: input %ux%u and output %ux%u differ (assert_eq! hsvdetector/imp.rs:121)",
                    in->width, in->height, out->width, out->height);
    if (((uint64_t)in->stride * in->height) % (uint64_t)*bpp != 0)
        return fail(MVFX_ERR_REFERENCE_PANIC, "hsvdetector: input plane size is not a multiple of %d bytes per pixel; the reference asserts on this (hsvdetector/imp.rs:122)", *bpp);
    return MVFX_OK;
}

// n frame pairs sharing geometry and formats (n == 1: the reference's transform_frame); <= kMaxBatch per launch.
int hsvdetector_impl(const mvfx_frame *ins, const mvfx_frame *outs, uint32_t n,
                     const mvfx_hsvdetector_settings *s, hipStream_t stream)
{
    if (!ins || !outs || !s || n == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector: NULL frame/settings or empty batch");
    const mvfx_frame *in = &ins[0], *out = &outs[0];
    int bpp, off;
    bool ibgr, a0, obgr;
    if (int rc = check_detect_pair(in, out, &bpp, &off, &ibgr, &a0, &obgr); rc != MVFX_OK) return rc;
    for (uint32_t i = 1; i < n; i++) {
        if (ins[i].width != in->width || ins[i].height != in->height || ins[i].stride != in->stride || ins[i].format != in->format ||
            outs[i].width != out->width || outs[i].height != out->height || outs[i].stride != out->stride || outs[i].format != out->format)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector: frames of one batch must share geometry and formats");
        if (int rc = check_packed_frame(&ins[i], "hsvdetector input"); rc != MVFX_OK) return rc;
        if (int rc = check_packed_frame(&outs[i], "hsvdetector output"); rc != MVFX_OK) return rc;
    }
    if (int rc = require_device(); rc != MVFX_OK)
        return rc;
    if (in->width == 0 || in->height == 0)
        return MVFX_OK;

    // |a - ref| <= -0.0  <=>  |a - ref| <= +0.0: canonicalise so the sign-based test sees +0
    const auto pz = [](float v) { return v == 0.0f ? 0.0f : v; };
    const HsvDetectorParams p{180.0f - s->hue_ref, pz(s->hue_var), s->saturation_ref, pz(s->saturation_var),
                              s->value_ref, pz(s->value_var), 180.0f, make_consts(nullptr)};
    const float dv[6] = {s->hue_ref, s->hue_var, s->saturation_ref, s->saturation_var, s->value_ref, s->value_var};
    bool det_fast_ok = std::fabs(p.ref_hue_offset) <= 360.0f;
    for (float f : dv) det_fast_ok = det_fast_ok && std::isfinite(f);
    // from_rgb's FAST form is settings-independent, so it is always valid here; the hue test has
    // its own domain (det_fast_ok).  variant option: 0 auto, 1 everything literal, 2 force both fast.
    const int g_variant = opt_hsv_variant();
    if (g_variant == 2 && !det_fast_ok)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector: settings are outside the proven domain of the strength-reduced hue test");
    const int variant = g_variant == 1 ? kGeneral : (det_fast_ok ? kDetFast : kFast);
    const uint64_t in_need = bpp == 4 ? 15 : 3;
    const bool flat = (uint64_t)in->width * bpp == in->stride && (uint64_t)out->width * 4 == out->stride;

    for (uint32_t done = 0; done < n; done += kMaxBatch) {
        const uint32_t m = (n - done) < (uint32_t)kMaxBatch ? (n - done) : (uint32_t)kMaxBatch;
        FrameBatch ifb{}, ofb{};
        uint64_t in_or = 0, out_or = 0;
        for (uint32_t i = 0; i < m; i++) {
            ifb.base[i] = static_cast<uint8_t *>(ins[done + i].data);
            ofb.base[i] = static_cast<uint8_t *>(outs[done + i].data);
            in_or |= (uint64_t)(uintptr_t)ins[done + i].data;
            out_or |= (uint64_t)(uintptr_t)outs[done + i].data;
        }
        uint64_t width = in->width, is = in->stride, os = out->stride;
        uint32_t rows = in->height;
        if (flat) {
            width = (uint64_t)in->width * in->height; rows = 1; is = 0; os = 0;
        } else {
            in_or |= is; out_or |= os;
        }
        const bool vec = (in_or & in_need) == 0 && (out_or & 15) == 0;
        const uint64_t work = vec ? (width + 3) / 4 : width;
        uint64_t bx = (work + kBlock - 1) / kBlock;
        if (bx > 65535u * 16u) bx = 65535u * 16u;
        const dim3 grid((uint32_t)bx, rows < 65535u ? rows : 65535u, m);
        const uint64_t in_bytes = (uint64_t)in->stride * in->height;
        const bool typed4 = opt_typed_loads() && vec && variant == kDetFast && bpp == 4 && (width & 3) == 0 && in_bytes < (1ull << 32);
        if (opt_direct_only() && !(typed4 && m == 1 && n == 1 && flat && completion_event() && (uint64_t)out->stride * out->height < (1ull << 32)))
            return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "hsvdetector: not a frame pair the direct-dispatch lane takes (unpadded 4-byte frames, strength-reduced settings, a completion event)");
        if (typed4) {
            const uint32_t iR = off + (ibgr ? 2 : 0), iG = off + 1, iB = off + (ibgr ? 0 : 2);
            const uint32_t word3 = (4 + iR) | ((4 + iG) << 3) | ((4 + iB) << 6) | (10u << 15); // see hsvfilter_impl
            const uint32_t o0 = obgr ? iB : iR, o2 = obgr ? iR : iB; // detect_px4_fast's selector, formed at run time
            const uint32_t sel = a0 ? (4u | (o0 << 8) | (iG << 16) | (o2 << 24)) : (o0 | (iG << 8) | (o2 << 16) | (4u << 24));
            // the direct-dispatch lane (direct_dispatch.h), as in hsvfilter_impl: one flat frame pair whose dependencies have finished or sit in front
            // of it on the lane queue it takes
            if (m == 1 && n == 1 && opt_direct() && flat && completion_event() && (uint64_t)out->stride * out->height < (1ull << 32)) {
                DirectDetArgs da{};
                da.in = ifb.base[0];
                da.out = ofb.base[0];
                da.groups = (uint32_t)(width / 4);
                da.word3 = word3;
                da.in_bytes = (uint32_t)in_bytes;
                da.perm_sel = sel;
                da.p = p;
                const int drc = direct_hsvdetector_submit(da, direct_queue_hint(stream));
                if (drc == MVFX_OK) continue;
                if (drc < 0) return drc;
            }
            if (opt_direct_only()) return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "hsvdetector: the direct-dispatch lane cannot take this frame");
            dim3 tgrid = grid; // MVFX_DET_TILE groups per lane
            tgrid.x = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((work + (uint64_t)kBlock * MVFX_DET_TILE - 1) / ((uint64_t)kBlock * MVFX_DET_TILE), 65535u * 16u));
            if (opt_nontemporal()) MVFX_LAUNCH(hsvdetector_typed_kernel<true>, tgrid, dim3(kBlock), 0, stream, ifb, ofb, width, rows, is, os, p, word3, (uint32_t)in_bytes, sel);
            else MVFX_LAUNCH(hsvdetector_typed_kernel<false>, tgrid, dim3(kBlock), 0, stream, ifb, ofb, width, rows, is, os, p, word3, (uint32_t)in_bytes, sel);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        if (opt_typed_loads() && vec && variant == kDetFast && bpp == 3 && (width & 3) == 0 && in_bytes < (1ull << 32)) {
            // 3-byte input by typed loads (hsvdetector3_typed_kernel).  Descriptor A delivers bytes 0, 1, 2 of the four fetched as R, G, B,
            // descriptor B bytes 1, 2, 3 (the lane's fourth pixel).  Colour selectors: the first byte of pixel j among the eight of its
            // (lo, hi) dword pair is 0, 3, 2, 1; the output keeps or swaps the outer two; 0x0c = constant zero for the alpha byte
            const uint32_t r0 = ibgr ? 2 : 0, b0 = ibgr ? 0 : 2;
            const uint32_t word3a = (4 + r0) | (5u << 3) | ((4 + b0) << 6) | (10u << 15), word3b = (5 + r0) | (6u << 3) | ((5 + b0) << 6) | (10u << 15);
            const uint32_t first[4] = {0, 3, 2, 1};
            uint32_t sels[4];
            for (int j = 0; j < 4; j++) {
                const uint32_t c0 = first[j], c1 = first[j] + 1, c2 = first[j] + 2;
                const uint32_t o0 = ibgr != obgr ? c2 : c0, o2 = ibgr != obgr ? c0 : c2;
                sels[j] = a0 ? (0x0cu | (o0 << 8) | (c1 << 16) | (o2 << 24)) : (o0 | (c1 << 8) | (o2 << 16) | (0x0cu << 24));
            }
            const uint4 sel4 = make_uint4(sels[0], sels[1], sels[2], sels[3]);
            const uint32_t alpha_mask = a0 ? 0x000000ffu : 0xff000000u;
            if (opt_nontemporal()) MVFX_LAUNCH(hsvdetector3_typed_kernel<true>, grid, dim3(kBlock), 0, stream, ifb, ofb, width, rows, is, os, p, word3a, word3b, (uint32_t)in_bytes, sel4, alpha_mask);
            else MVFX_LAUNCH(hsvdetector3_typed_kernel<false>, grid, dim3(kBlock), 0, stream, ifb, ofb, width, rows, is, os, p, word3a, word3b, (uint32_t)in_bytes, sel4, alpha_mask);
            MVFX_HIP_TRY(hipGetLastError());
            continue;
        }
        if (vec) {
            if (variant == kDetFast) launch_detect<kDetFast, kModeVec4>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
            else if (variant == kFast) launch_detect<kFast, kModeVec4>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
            else launch_detect<kGeneral, kModeVec4>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
        } else {
            if (variant == kDetFast) launch_detect<kDetFast, kModeBytes>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
            else if (variant == kFast) launch_detect<kFast, kModeBytes>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
            else launch_detect<kGeneral, kModeBytes>(bpp, off, ibgr, a0, obgr, grid, stream, ifb, ofb, width, rows, is, os, p);
        }
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}


int hsvdetector_i420_impl(const mvfx_planar_frame *in, const mvfx_frame *out, const mvfx_hsvdetector_settings *s, int yuv_standard,
                          hipStream_t stream)
{
    if (!in || !out || !s)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector_i420: NULL frame or settings");
    if (in->format != MVFX_FORMAT_I420)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvdetector_i420: input must be I420");
    bool a0, obgr;
    if (detect_out_layout(out->format, &a0, &obgr) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsvdetector: output format %d not in RGBA ARGB BGRA ABGR (hsvdetector/imp.rs:89-96)", out->format);
    if (int rc = check_packed_frame(out, "hsvdetector output"); rc != MVFX_OK) return rc;
    if (in->width != out->width || in->height != out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "hsvdetector: input %ux%u and output %ux%u differ (assert_eq! hsvdetector/imp.rs:121)",
                    in->width, in->height, out->width, out->height);
    if (yuv_standard < 0 || yuv_standard > 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector_i420: yuv_standard %d is not 0..3", yuv_standard);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t w = in->width, h = in->height, cw = (w + 1) / 2;
    if (w == 0 || h == 0) return MVFX_OK;
    for (int pidx = 0; pidx < 3; pidx++)
        if (!in->data[pidx] || in->stride[pidx] < (pidx == 0 ? w : cw))
            return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector_i420: bad plane %d", pidx);
    if ((h + 1) / 2 > 65535u)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector_i420: height %u too large", h);
    const auto pz = [](float v) { return v == 0.0f ? 0.0f : v; };
    const HsvDetectorParams p{180.0f - s->hue_ref, pz(s->hue_var), s->saturation_ref, pz(s->saturation_var),
                              s->value_ref, pz(s->value_var), 180.0f, make_consts(nullptr)};
    const float dv[6] = {s->hue_ref, s->hue_var, s->saturation_ref, s->saturation_var, s->value_ref, s->value_var};
    bool det_fast_ok = std::fabs(p.ref_hue_offset) <= 360.0f;
    for (float f : dv) det_fast_ok = det_fast_ok && std::isfinite(f);
    const int g_variant = opt_hsv_variant();
    if (g_variant == 2 && !det_fast_ok)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector: settings are outside the proven domain of the strength-reduced hue test");
    const int variant = g_variant == 1 ? kGeneral : (det_fast_ok ? kDetFast : kFast);
    const YuvToRgbCoef k = yuv_to_rgb_coef(pick_yuv_standard(h, yuv_standard));
    const uint8_t *yp = static_cast<const uint8_t *>(in->data[0]), *up = static_cast<const uint8_t *>(in->data[1]), *vp = static_cast<const uint8_t *>(in->data[2]);
    const bool aligned = ((reinterpret_cast<uintptr_t>(yp) | in->stride[0]) & 7) == 0 &&
                         ((reinterpret_cast<uintptr_t>(up) | in->stride[1] | reinterpret_cast<uintptr_t>(vp) | in->stride[2]) & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(out->data) | out->stride) & 15) == 0;
    // eight columns per lane: unlike the plain converter this kernel is VALU bound (the detector's arithmetic), four columns with
    // their contiguous stores change nothing (4K: 20.2 vs 20.7 us)
    constexpr int kCols = 8;
    const dim3 grid(((w + kCols - 1) / kCols + kBlock - 1) / kBlock, (h + 1) / 2);
    uint8_t *o = static_cast<uint8_t *>(out->data);
#define MVFX_DI(A, B, V, AL) \
    MVFX_LAUNCH((hsvdetector_i420_kernel<A, B, V, AL, kCols>), grid, dim3(kBlock), 0, stream, yp, up, vp, (uint64_t)in->stride[0], \
                       (uint64_t)in->stride[1], (uint64_t)in->stride[2], w, h, k, p, o, (uint64_t)out->stride)
#define MVFX_DI_V(A, B, AL) \
    do { if (variant == kDetFast) MVFX_DI(A, B, kDetFast, AL); else if (variant == kFast) MVFX_DI(A, B, kFast, AL); else MVFX_DI(A, B, kGeneral, AL); } while (0)
#define MVFX_DI_AL(A, B) \
    do { if (aligned) MVFX_DI_V(A, B, true); else MVFX_DI_V(A, B, false); } while (0)
    if (a0) { if (obgr) MVFX_DI_AL(true, true); else MVFX_DI_AL(true, false); }
    else { if (obgr) MVFX_DI_AL(false, true); else MVFX_DI_AL(false, false); }
#undef MVFX_DI_AL
#undef MVFX_DI_V
#undef MVFX_DI
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_hsvdetector_transform_i420(const mvfx_planar_frame *i420_in, const mvfx_frame *out_frame,
                                    const mvfx_hsvdetector_settings *settings, int32_t yuv_standard, mvfx_stream stream)
{
    return hsvdetector_i420_impl(i420_in, out_frame, settings, yuv_standard, as_stream(stream));
}

int mvfx_hsvfilter_transform_frame_ip(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings,
                                      mvfx_stream stream)
{
    return hsvfilter_impl(frame, 1, settings, as_stream(stream));
}

int mvfx_hsvfilter_transform_frames_ip(const mvfx_frame *frames, uint32_t n_frames,
                                       const mvfx_hsvfilter_settings *settings, mvfx_stream stream)
{
    return hsvfilter_impl(frames, n_frames, settings, as_stream(stream));
}

int mvfx_hsvfilter_transform_frames_ip_settings(const mvfx_frame *frames, uint32_t n_frames,
                                                const mvfx_hsvfilter_settings *settings, mvfx_stream stream)
{
    return hsvfilter_frames_impl(frames, n_frames, settings, as_stream(stream));
}

int mvfx_hsvfilter_transform_i420(const mvfx_planar_frame *i420_in, const mvfx_planar_frame *i420_out,
                                  const mvfx_hsvfilter_settings *settings, int32_t yuv_standard, mvfx_stream stream)
{
    return hsvfilter_i420_impl(i420_in, i420_out, settings, yuv_standard, as_stream(stream));
}

int mvfx_hsvfilter_transform_frame_ip_host(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings)
{
    if (!frame || !settings)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: NULL frame or settings");
    if (int rc = check_packed_frame(frame, "hsvfilter"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t bytes = (size_t)frame->stride * frame->height;
    if (bytes == 0)
        return MVFX_OK;
    void *dev = nullptr;
    if (int rc = host_scratch(bytes, 0, &dev); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    MVFX_HIP_TRY(hipMemcpyAsync(dev, frame->data, bytes, hipMemcpyHostToDevice, st));
    mvfx_frame d = *frame;
    d.data = dev;
    if (int rc = hsvfilter_impl(&d, 1, settings, st); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(frame->data, dev, bytes, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return MVFX_OK;
}

int mvfx_hsvdetector_transform_frame(const mvfx_frame *in_frame, const mvfx_frame *out_frame,
                                     const mvfx_hsvdetector_settings *settings, mvfx_stream stream)
{
    return hsvdetector_impl(in_frame, out_frame, 1, settings, as_stream(stream));
}

int mvfx_hsvdetector_transform_frames(const mvfx_frame *in_frames, const mvfx_frame *out_frames, uint32_t n_frames,
                                      const mvfx_hsvdetector_settings *settings, mvfx_stream stream)
{
    return hsvdetector_impl(in_frames, out_frames, n_frames, settings, as_stream(stream));
}

int mvfx_hsvdetector_transform_frame_host(const mvfx_frame *in_frame, const mvfx_frame *out_frame,
                                          const mvfx_hsvdetector_settings *settings)
{
    if (!in_frame || !out_frame || !settings)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvdetector: NULL frame or settings");
    if (int rc = check_packed_frame(in_frame, "hsvdetector input"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(out_frame, "hsvdetector output"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t ib = (size_t)in_frame->stride * in_frame->height;
    const size_t ob = (size_t)out_frame->stride * out_frame->height;
    if (ib == 0 || ob == 0)
        return hsvdetector_impl(in_frame, out_frame, 1, settings, nullptr);
    void *din = nullptr, *dout = nullptr;
    if (int rc = host_scratch(ib, 0, &din); rc != MVFX_OK) return rc;
    if (int rc = host_scratch(ob, 1, &dout); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    MVFX_HIP_TRY(hipMemcpyAsync(din, in_frame->data, ib, hipMemcpyHostToDevice, st));
    // row padding of the output buffer is not written by the kernel: keep the caller's bytes
    if ((size_t)out_frame->width * 4 != out_frame->stride)
        MVFX_HIP_TRY(hipMemcpyAsync(dout, out_frame->data, ob, hipMemcpyHostToDevice, st));
    mvfx_frame di = *in_frame, dof = *out_frame;
    di.data = din;
    dof.data = dout;
    if (int rc = hsvdetector_impl(&di, &dof, 1, settings, st); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(out_frame->data, dout, ob, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return MVFX_OK;
}

int mvfx_hsv_from_frame(const mvfx_frame *frame, float *hsv_out_device, mvfx_stream stream)
{
    if (!frame || !hsv_out_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsv_from_frame: NULL argument");
    int bpp, off;
    bool bgr;
    if (filter_layout(frame->format, &bpp, &off, &bgr) != 0 || bpp != 4)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "hsv_from_frame: needs a 4-byte packed format");
    if (int rc = check_packed_frame(frame, "hsv_from_frame"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (frame->width == 0 || frame->height == 0)
        return MVFX_OK;
    const dim3 grid((frame->width + kBlock - 1) / kBlock, frame->height < 65535u ? frame->height : 65535u, 1);
    const uint8_t *in = static_cast<const uint8_t *>(frame->data);
    hipStream_t st = as_stream(stream);
    const bool fast = opt_hsv_variant() != 1;
    const FastConsts kc = make_consts(nullptr);
#define MVFX_LH(O, B) \
    do { if (fast) MVFX_LAUNCH((hsv_from_frame_kernel<O, B, kFast>), grid, dim3(kBlock), 0, st, in, hsv_out_device, frame->width, frame->height, (uint64_t)frame->stride, kc); \
         else MVFX_LAUNCH((hsv_from_frame_kernel<O, B, kGeneral>), grid, dim3(kBlock), 0, st, in, hsv_out_device, frame->width, frame->height, (uint64_t)frame->stride, kc); } while (0)
    if (off == 0) { if (bgr) MVFX_LH(0, true); else MVFX_LH(0, false); }
    else { if (bgr) MVFX_LH(1, true); else MVFX_LH(1, false); }
#undef MVFX_LH
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

} // extern "C"
