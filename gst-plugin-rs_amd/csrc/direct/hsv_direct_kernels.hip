// The kernels of the direct-dispatch lane (csrc/direct_dispatch.h): hsvfilter on ONE flat packed 4-byte frame, typed loads, two 16-byte pixel
// groups per lane, one-shot grid -- hsvfilter4_typed_kernel<kFast | kFastNeg, 2, NT> of csrc/hsv_typed_kernels.hip with the frame loop taken out
// and a kernel argument block of 112 bytes instead of the batch's 32 frame pointers.  Same loads, same pixel function (hsv_math.hpp); the
// stores are write-through (below): same bytes (tests/test_direct_dispatch_gpu.py: all 2^24 colours).  Built into a bare code object (gst-plugin-rs_amd/Makefile:
// --genco --no-gpu-bundle-output) that the library embeds and loads through the HSA executable API; extern "C" names, no gridDim / blockDim
// (implicit kernel arguments), 256 lanes per workgroup.
#include "direct_dispatch.h"
#include "hsv_filter_lds.hpp"

namespace mvfx {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The stores are WRITE-THROUGH at system scope (sc0 sc1) and drained before the wave ends: the lane's packets carry NO release fence (an L2
// write-back walk on eight XCDs per dispatch, 1.2 us of every 4K frame: profiles/r6/aql_probe_real_kernel_store_policies.txt), so the bytes must be
// in memory when the completion signal fires -- the write-through publish of MI355X_MICROARCH.md ("sc1 stores + vmcnt(0)"); whoever reads them next
// does its own acquire.  NT (MVFX_OPT_NONTEMPORAL: nobody on the device reads the frame next): the stores carry the non-temporal hint as well.
template <int VARIANT, bool NT>
__device__ __forceinline__ void direct_hsvfilter4(const DirectHsvArgs &a)
{
    constexpr int kBlock = kHsvBlock, TILE = 2;
    __shared__ FilterLds lds;
    init_filter_lds<VARIANT>(lds, a.off, a.bgr != 0);
    const uint64_t base = reinterpret_cast<uint64_t>(a.frame);
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) & 0xffffu)); // stride 0: raw byte offsets
    rs.z = __builtin_amdgcn_readfirstlane((int)a.frame_bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)a.word3);
    const uint32_t g0 = blockIdx.x * (uint32_t)(kBlock * TILE) + threadIdx.x;
    u32x4 raw[TILE];
    f32x3 c[TILE][4];
    uint32_t voff[TILE];
#pragma unroll
    for (int u = 0; u < TILE; u++) // groups past the end: the buffer bounds check returns zeros, nothing is stored
        voff[u] = (g0 + (uint32_t)u * kBlock) << 4;
    // all loads of a lane and their wait are ONE asm statement (hsv_typed_kernels.hip: the compiler does not track asm loads)
#define MVFX_DIRECT_LOADS2(NTS)                                                                \
    asm volatile("buffer_load_dwordx4 %0, %10, %12, 0 offen" NTS "\n\t"                          \
                 "buffer_load_dwordx4 %1, %11, %12, 0 offen" NTS "\n\t"                          \
                 "buffer_load_format_xyz %2, %10, %12, 0 offen\n\t"                              \
                 "buffer_load_format_xyz %3, %10, %12, 0 offen offset:4\n\t"                     \
                 "buffer_load_format_xyz %4, %10, %12, 0 offen offset:8\n\t"                     \
                 "buffer_load_format_xyz %5, %10, %12, 0 offen offset:12\n\t"                    \
                 "buffer_load_format_xyz %6, %11, %12, 0 offen\n\t"                              \
                 "buffer_load_format_xyz %7, %11, %12, 0 offen offset:4\n\t"                     \
                 "buffer_load_format_xyz %8, %11, %12, 0 offen offset:8\n\t"                     \
                 "buffer_load_format_xyz %9, %11, %12, 0 offen offset:12\n\t"                    \
                 "s_waitcnt vmcnt(0)"                                                             \
                 : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]), \
                   "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[1][2]), "=&v"(c[1][3])                 \
                 : "v"(voff[0]), "v"(voff[1]), "s"(rs)                                            \
                 : "memory")
    MVFX_DIRECT_LOADS2(""); // (cached loads in both forms: `nt` loads lose 3 % here, profiles/r6/direct_lane_one_thread.txt)
#undef MVFX_DIRECT_LOADS2
#pragma unroll
    for (int u = 0; u < TILE; u++) {
        const uint32_t g = g0 + (uint32_t)u * kBlock;
        if (g < a.groups) {
            uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t T;
                const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(c[u][j].x, c[u][j].y, c[u][j].z, a.p, T);
                w[j] = __builtin_amdgcn_perm(T, w[j], sextant_at(lds.sextant, sel_off));
            }
            const u32x4 t = {w[0], w[1], w[2], w[3]};
            u32x4 *dst = reinterpret_cast<u32x4 *>(a.frame + ((uint64_t)g << 4));
            // (s_nop: a VMEM store of more than 64 bits reads its data registers up to two wait states after issue, and the compiler -- which
            // inserts those wait states behind its own stores -- does not know this asm is one: without them ~0.2 % of the pixels came out wrong)
            // NT (MVFX_OPT_NONTEMPORAL: nobody on the device reads the frame next): the non-temporal hint on top, as csrc/device_store.hpp
            if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store of this wave acknowledged (the compiler does not count asm stores)
}


// hsvdetector (hsvdetector_typed_kernel of hsv_kernels.hip, one flat frame pair, one-shot grid): u8 / 255 by typed loads, the strength-reduced hue
// test, the output pixel by one v_perm_b32 from the raw input dword and the hit mask (hsvdetector/imp.rs:100-160); write-through stores.
__device__ __forceinline__ void direct_hsvdetector4(const DirectDetArgs &a)
{
    constexpr int kBlock = kHsvBlock, TILE = 2;
    const uint64_t base = reinterpret_cast<uint64_t>(a.in);
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)a.in_bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)a.word3);
    const uint32_t sel = __builtin_amdgcn_readfirstlane(a.perm_sel);
    const uint32_t g0 = blockIdx.x * (uint32_t)(kBlock * TILE) + threadIdx.x;
    u32x4 raw[TILE];
    f32x3 c[TILE][4];
    uint32_t voff[TILE];
#pragma unroll
    for (int u = 0; u < TILE; u++) voff[u] = (g0 + (uint32_t)u * kBlock) << 4;
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
                 : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]),
                   "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[1][2]), "=&v"(c[1][3])
                 : "v"(voff[0]), "v"(voff[1]), "s"(rs)
                 : "memory");
#pragma unroll
    for (int u = 0; u < TILE; u++) {
        const uint32_t g = g0 + (uint32_t)u * kBlock;
        if (g < a.groups) {
            const uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const HsvN hsv = from_unit_rgb_fast_n(c[u][j].x, c[u][j].y, c[u][j].z, a.p.consts);
                r[j] = __builtin_amdgcn_perm(~detect_miss_mask_fast(hsv, a.p), w[j], sel); // selector byte 4 = the hit mask
            }
            const u32x4 t = {r[0], r[1], r[2], r[3]};
            u32x4 *dst = reinterpret_cast<u32x4 *>(a.out + ((uint64_t)g << 4));
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

} // namespace
} // namespace mvfx

extern "C" __global__ __launch_bounds__(256) void mvfx_direct_hsvfilter4_pos(mvfx::DirectHsvArgs a) { mvfx::direct_hsvfilter4<mvfx::kFast, false>(a); }
extern "C" __global__ __launch_bounds__(256) void mvfx_direct_hsvfilter4_pos_nt(mvfx::DirectHsvArgs a) { mvfx::direct_hsvfilter4<mvfx::kFast, true>(a); }
extern "C" __global__ __launch_bounds__(256) void mvfx_direct_hsvfilter4_neg(mvfx::DirectHsvArgs a) { mvfx::direct_hsvfilter4<mvfx::kFastNeg, false>(a); }
extern "C" __global__ __launch_bounds__(256) void mvfx_direct_hsvfilter4_neg_nt(mvfx::DirectHsvArgs a) { mvfx::direct_hsvfilter4<mvfx::kFastNeg, true>(a); }
extern "C" __global__ __launch_bounds__(256) void mvfx_direct_hsvdetector4(mvfx::DirectDetArgs a) { mvfx::direct_hsvdetector4(a); }
