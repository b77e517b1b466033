// hsvfilter4_typed_kernel: hsvfilter on 4-byte formats with the three `byte / 255.0` divisions of every pixel done by
// the texture unit (video/hsv/src/hsvutils.rs:45-55 `let r = in_p[0] as f32 / 255.0` ...).  Own translation unit: built
// with the max-memory-clause scheduling strategy (gst-plugin-rs_amd/Makefile HSVTYPED_EXTRA: 88.2 k fps against 87.6 k with
// LLVM's default GCN scheduler and 86.2 k with the iterative-ilp strategy hsv_kernels.hip is built with).
#include "hsv_filter_lds.hpp"
#include "device_store.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace mvfx {
namespace {

constexpr int kBlock = kHsvBlock;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A typed buffer load (buffer_load_format_xyz, DATA_FORMAT 8_8_8_8, NUM_FORMAT UNORM) returns RN(byte / 255.0f) for
// every byte value in every channel (tools/probe_unorm.hip: 256 x 4 values, all exact), and the descriptor's DST_SEL
// hands the channels over as (R, G, B) whatever the byte order: the three exact divisions of a pixel -- 9 of its 63
// VALU instructions (v_cvt_f32_ubyteN + v_mul + v_fmac each) -- are done by the memory pipeline on the way in.
// The raw dwords (4th byte, v_perm source) come from an untyped buffer_load_dwordx4 through the same descriptor; the
// typed loads of the same 16 bytes follow immediately and merge into its cache lines.  All loads of a lane and their
// wait are ONE asm statement: the compiler does not track asm loads, so nothing may touch their registers in between.
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT, int TILE, bool NT>
__device__ __forceinline__ void hsvfilter4_typed_body(const FrameBatch &fb, uint64_t width, uint32_t rows, uint64_t stride,
                                                      const FastConsts &p, uint32_t word3, uint32_t frame_bytes, int off, bool bgr)
{
    static_assert(VARIANT == kFast || VARIANT == kFastNeg, "strength-reduced variants only");
    __shared__ FilterLds lds;
    init_filter_lds<VARIANT>(lds, off, bgr);
    uint8_t *frame = fb.base[blockIdx.z];
    const uint64_t a = reinterpret_cast<uint64_t>(frame);
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)); // stride 0: raw byte offsets
    rs.z = __builtin_amdgcn_readfirstlane((int)frame_bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)word3);
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *line = frame + (uint64_t)row * stride;
        const uint32_t line_off = (uint32_t)((uint64_t)row * stride);
        const uint64_t groups = width >> 2; // the launcher guarantees width % 4 == 0
        // (one-shot grid, one chunk per workgroup: workgroups walking 2 / 3 / 4 / 8 adjacent chunks, capped grids and one group per lane were measured
        // for the one-frame launch and all lose 2-15 %: profiles/r6/single_frame_launch_shapes.txt)
        for (uint64_t t0 = (uint64_t)blockIdx.x * (kBlock * TILE); t0 < groups; t0 += (uint64_t)gridDim.x * (kBlock * TILE)) {
            u32x4 raw[TILE];
            f32x3 c[TILE][4];
            uint32_t voff[TILE];
#pragma unroll
            for (int u = 0; u < TILE; u++) // groups past the end: the buffer bounds check returns zeros, nothing is stored
                voff[u] = line_off + (uint32_t)((t0 + (uint64_t)u * kBlock + threadIdx.x) << 4);
#define MVFX_TYPED_LOADS2(NTS)                                                                 \
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
#define MVFX_TYPED_LOADS1(NTS)                                                                 \
    asm volatile("buffer_load_dwordx4 %0, %5, %6, 0 offen" NTS "\n\t"                            \
                 "buffer_load_format_xyz %1, %5, %6, 0 offen\n\t"                                \
                 "buffer_load_format_xyz %2, %5, %6, 0 offen offset:4\n\t"                       \
                 "buffer_load_format_xyz %3, %5, %6, 0 offen offset:8\n\t"                       \
                 "buffer_load_format_xyz %4, %5, %6, 0 offen offset:12\n\t"                      \
                 "s_waitcnt vmcnt(0)"                                                             \
                 : "=&v"(raw[0]), "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3])  \
                 : "v"(voff[0]), "s"(rs)                                                          \
                 : "memory")
            if constexpr (TILE == 2) {
                if constexpr (NT && MVFX_STREAM_NT_LOADS) MVFX_TYPED_LOADS2(" nt"); else MVFX_TYPED_LOADS2("");
            } else {
                if constexpr (NT && MVFX_STREAM_NT_LOADS) MVFX_TYPED_LOADS1(" nt"); else MVFX_TYPED_LOADS1("");
            }
#undef MVFX_TYPED_LOADS2
#undef MVFX_TYPED_LOADS1
#pragma unroll
            for (int u = 0; u < TILE; u++) {
                const uint64_t g = t0 + (uint64_t)u * kBlock + threadIdx.x;
                if (g < groups) {
                    uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t T;
                        const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(c[u][j].x, c[u][j].y, c[u][j].z, p, T);
                        w[j] = __builtin_amdgcn_perm(T, w[j], sextant_at(lds.sextant, sel_off));
                    }
                    const u32x4 t = {w[0], w[1], w[2], w[3]};
                    stream_store16<NT>(line + (g << 4), t);
                }
            }
        }
    }
}

template <int VARIANT, int TILE, bool NT>
__global__ __launch_bounds__(kBlock) void hsvfilter4_typed_kernel(FrameBatch fb, uint64_t width, uint32_t rows, uint64_t stride,
                                                                  FastConsts p, uint32_t word3, uint32_t frame_bytes, int off, bool bgr)
{
    hsvfilter4_typed_body<VARIANT, TILE, NT>(fb, width, rows, stride, p, word3, frame_bytes, off, bgr);
}

// The same kernel with the settings of every frame of the launch in the argument block (blockIdx.z = frame = stream): what the
// launch combiner needs -- the elements of a process each keep their own `hue-shift` ... properties (hsvfilter/imp.rs:32-39), yet
// their frames share one launch.  The per-frame values are wave-uniform scalar loads from the kernel arguments.
template <int VARIANT, int TILE, bool NT>
__global__ __launch_bounds__(kBlock) void hsvfilter4_typed_frames_kernel(FrameBatch fb, uint64_t width, uint32_t rows, uint64_t stride,
                                                                         FastConsts p, FrameSettingsBatch fs, uint32_t word3,
                                                                         uint32_t frame_bytes, int off, bool bgr)
{
    const FrameSettings &f = fs.s[blockIdx.z];
    p.hue_shift = f.hue_shift;
    p.saturation_mul = f.saturation_mul;
    p.saturation_off = f.saturation_off;
    p.value_mul = f.value_mul;
    p.value_off = f.value_off;
    p.neg_saturation_mul = f.neg_saturation_mul;
    hsvfilter4_typed_body<VARIANT, TILE, NT>(fb, width, rows, stride, p, word3, frame_bytes, off, bgr);
}


// ---- 3-byte formats (RGB / BGR) with typed loads (round 5) -----------------------------------------------------------------------
// A typed buffer load of DATA_FORMAT 8_8_8_8 does not need a 4-byte aligned address on gfx950 (tools/probes/typed_unaligned.hip: byte
// offsets 1, 2, 3 x lane, all 1 024 channel values exact), so a packed RGB pixel at byte 3 i is fetched like an RGBA one: four bytes
// from 3 i, the first three delivered as RN(byte / 255).  The VALU kernel (hsvfilter3_kernel) spends 13 of its 67 instructions per pixel
// on taking the twelve bytes of a lane apart and dividing them; here the texture unit does both.  A lane owns 12 bytes = 4 pixels and
// never reads outside them: pixels 0..2 from byte offsets 0, 3, 6 through descriptor A (DST_SEL x, y, z = bytes 0, 1, 2), pixel 3 from
// byte offset 8 through descriptor B (bytes 1, 2, 3) -- the last pixel of a frame is read without touching the byte behind it.  The four
// result dwords [c0 c1 c2 .] are packed into three with three v_perm_b32.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <int VARIANT, int TILE, bool NT>
__global__ __launch_bounds__(kBlock) void hsvfilter3_typed_kernel(FrameBatch fb, uint64_t width, uint32_t rows, uint64_t stride, FastConsts p,
                                                                  uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, bool bgr)
{
    static_assert(VARIANT == kFast || VARIANT == kFastNeg, "strength-reduced variants only");
    __shared__ FilterLds lds;
    init_filter_lds<VARIANT>(lds, 0, bgr);
    uint8_t *frame = fb.base[blockIdx.z];
    const uint64_t a = reinterpret_cast<uint64_t>(frame);
    i32x4 ra, rb;
    ra.x = rb.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    ra.y = rb.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)); // stride 0: raw byte offsets
    ra.z = rb.z = __builtin_amdgcn_readfirstlane((int)frame_bytes);
    ra.w = __builtin_amdgcn_readfirstlane((int)word3a);
    rb.w = __builtin_amdgcn_readfirstlane((int)word3b);
    for (uint32_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint8_t *line = frame + (uint64_t)row * stride;
        const uint32_t line_off = (uint32_t)((uint64_t)row * stride);
        const uint64_t groups = width >> 2; // the launcher guarantees width % 4 == 0 (other widths: hsvfilter3_typed_rows_kernel)
        for (uint64_t t0 = (uint64_t)blockIdx.x * (kBlock * TILE); t0 < groups; t0 += (uint64_t)gridDim.x * (kBlock * TILE)) {
            f32x3 c[TILE][4];
            uint32_t voff[TILE];
#pragma unroll
            for (int u = 0; u < TILE; u++) // groups past the end: the buffer bounds check returns zeros, nothing is stored
                voff[u] = line_off + (uint32_t)(t0 + (uint64_t)u * kBlock + threadIdx.x) * 12u;
            // all loads of a lane and their wait are ONE asm statement: the compiler does not track asm loads, so nothing may touch their
            // registers in between (hsvfilter4_typed_body)
#define MVFX_TYPED3_2(NTS)                                                                        \
    asm volatile("buffer_load_format_xyz %0, %8, %10, 0 offen" NTS "\n\t"                          \
                 "buffer_load_format_xyz %1, %8, %10, 0 offen offset:3" NTS "\n\t"                 \
                 "buffer_load_format_xyz %2, %8, %10, 0 offen offset:6" NTS "\n\t"                 \
                 "buffer_load_format_xyz %3, %8, %11, 0 offen offset:8" NTS "\n\t"                 \
                 "buffer_load_format_xyz %4, %9, %10, 0 offen" NTS "\n\t"                          \
                 "buffer_load_format_xyz %5, %9, %10, 0 offen offset:3" NTS "\n\t"                 \
                 "buffer_load_format_xyz %6, %9, %10, 0 offen offset:6" NTS "\n\t"                 \
                 "buffer_load_format_xyz %7, %9, %11, 0 offen offset:8" NTS "\n\t"                 \
                 "s_waitcnt vmcnt(0)"                                                               \
                 : "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]), "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[1][2]), "=&v"(c[1][3]) \
                 : "v"(voff[0]), "v"(voff[1]), "s"(ra), "s"(rb)                                     \
                 : "memory")
#define MVFX_TYPED3_1(NTS)                                                                        \
    asm volatile("buffer_load_format_xyz %0, %4, %5, 0 offen" NTS "\n\t"                           \
                 "buffer_load_format_xyz %1, %4, %5, 0 offen offset:3" NTS "\n\t"                  \
                 "buffer_load_format_xyz %2, %4, %5, 0 offen offset:6" NTS "\n\t"                  \
                 "buffer_load_format_xyz %3, %4, %6, 0 offen offset:8" NTS "\n\t"                  \
                 "s_waitcnt vmcnt(0)"                                                               \
                 : "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3])                   \
                 : "v"(voff[0]), "s"(ra), "s"(rb)                                                   \
                 : "memory")
            if constexpr (TILE == 2) {
                if constexpr (NT && MVFX_STREAM_NT_LOADS) MVFX_TYPED3_2(" nt"); else MVFX_TYPED3_2("");
            } else {
                if constexpr (NT && MVFX_STREAM_NT_LOADS) MVFX_TYPED3_1(" nt"); else MVFX_TYPED3_1("");
            }
#undef MVFX_TYPED3_2
#undef MVFX_TYPED3_1
#pragma unroll
            for (int u = 0; u < TILE; u++) {
                const uint64_t g = t0 + (uint64_t)u * kBlock + threadIdx.x;
                if (g < groups) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t T;
                        const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(c[u][j].x, c[u][j].y, c[u][j].z, p, T);
                        w[j] = __builtin_amdgcn_perm(T, 0u, sextant_at(lds.sextant, sel_off)); // [c0 c1 c2 0] in memory order
                    }
                    // twelve bytes: p0 = d0[0..2], p1 = d0[3] d1[0..1], p2 = d1[2..3] d2[0], p3 = d2[1..3]   (v_perm: bytes 0-3 = S1, 4-7 = S0)
                    const u32x3 t = {__builtin_amdgcn_perm(w[1], w[0], 0x04020100u), __builtin_amdgcn_perm(w[2], w[1], 0x05040201u),
                                     __builtin_amdgcn_perm(w[3], w[2], 0x06050402u)};
                    stream_store12<NT>(line + g * 12, t);
                }
            }
        }
    }
}


// ---- self-test of the hardware property the 3-byte kernels rest on ----------------------------------------------------------------
// Lane i issues exactly the four loads of a lane of hsvfilter3_typed_kernel / hsvdetector3_typed_kernel -- descriptor A at immediate
// offsets 0, 3, 6, descriptor B at 8 -- from byte address i, so the 1 024 lanes cover every address alignment (i mod 4) with every byte
// value in every channel (the host fills position 4 k + r with a bijection of k).  Twelve floats per lane go back for the host to compare
// with RN(byte / 255).
__global__ __launch_bounds__(kBlock) void typed_unorm8_selftest_kernel(const uint8_t *bytes, float *out, uint32_t n_bytes, uint32_t word3a,
                                                                       uint32_t word3b, uint32_t lanes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(bytes);
    i32x4 ra, rb;
    ra.x = rb.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    ra.y = rb.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    ra.z = rb.z = __builtin_amdgcn_readfirstlane((int)n_bytes);
    ra.w = __builtin_amdgcn_readfirstlane((int)word3a);
    rb.w = __builtin_amdgcn_readfirstlane((int)word3b);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    f32x3 c[4];
    asm volatile("buffer_load_format_xyz %0, %4, %5, 0 offen\n\t"
                 "buffer_load_format_xyz %1, %4, %5, 0 offen offset:3\n\t"
                 "buffer_load_format_xyz %2, %4, %5, 0 offen offset:6\n\t"
                 "buffer_load_format_xyz %3, %4, %6, 0 offen offset:8\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "v"(i), "s"(ra), "s"(rb)
                 : "memory");
    if (i < lanes) {
        float *o = out + (size_t)i * 12;
#pragma unroll
        for (int j = 0; j < 4; j++) { o[3 * j] = c[j].x; o[3 * j + 1] = c[j].y; o[3 * j + 2] = c[j].z; }
    }
}

// RGB / BGR frames whose width is NOT a multiple of four (854, 1366 ... : round 6).  Such a frame always has row padding (the
// stride is a multiple of four, three times such a width is not), so it cannot be walked as one run of pixels, and walking it row by row leaves
// most lanes of the last workgroup of every row idle (1366 wide: 341 groups = 1.33 workgroups per row).  Here the WHOLE frame's groups are one
// index space: lane L takes group (L / gpr, L mod gpr) with the division by a precomputed reciprocal (exact for L x gpr < 2^32: the launcher
// checks), two groups per lane, the typed loads of hsvfilter3_typed_kernel.  Workgroups with blockIdx.y == 1 serve the rows' last 1..3 pixels,
// one lane per pixel: the same typed load -- four bytes from the pixel's first byte, of which the row's padding supplies the fourth -- and three
// byte stores.
template <int VARIANT, bool NT>
__global__ __launch_bounds__(kBlock) void hsvfilter3_typed_rows_kernel(FrameBatch fb, uint32_t width, uint32_t rows, uint32_t stride, FastConsts p, uint32_t word3a,
                                                                       uint32_t word3b, uint32_t frame_bytes, bool bgr, uint32_t gpr, uint32_t gpr_magic)
{
    constexpr int TILE = 2;
    __shared__ FilterLds lds;
    init_filter_lds<VARIANT>(lds, 0, bgr);
    uint8_t *frame = fb.base[blockIdx.z];
    const uint64_t a = reinterpret_cast<uint64_t>(frame);
    i32x4 ra, rb;
    ra.x = rb.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    ra.y = rb.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    ra.z = rb.z = __builtin_amdgcn_readfirstlane((int)frame_bytes);
    ra.w = __builtin_amdgcn_readfirstlane((int)word3a);
    rb.w = __builtin_amdgcn_readfirstlane((int)word3b);
    if (blockIdx.y == 0) {
        const uint32_t total = rows * gpr;
        for (uint32_t t0 = blockIdx.x * (uint32_t)(kBlock * TILE); t0 < total; t0 += gridDim.x * (uint32_t)(kBlock * TILE)) {
            f32x3 c[TILE][4];
            uint32_t voff[TILE], at[TILE];
            bool valid[TILE];
#pragma unroll
            for (int u = 0; u < TILE; u++) {
                const uint32_t L = t0 + (uint32_t)u * kBlock + threadIdx.x;
                valid[u] = L < total;
                const uint32_t Lc = valid[u] ? L : 0u;
                const uint32_t row = gpr_magic ? __umulhi(Lc, gpr_magic) : Lc, col = Lc - row * gpr; // (magic 0: one group per row)
                at[u] = row * stride + col * 12u;
                voff[u] = valid[u] ? at[u] : frame_bytes; // past the end: the bounds check returns zeros, nothing is stored
            }
#define MVFX_TYPED3R(NTS)                                                                        \
    asm volatile("buffer_load_format_xyz %0, %8, %10, 0 offen" NTS "\n\t"                          \
                 "buffer_load_format_xyz %1, %8, %10, 0 offen offset:3" NTS "\n\t"                 \
                 "buffer_load_format_xyz %2, %8, %10, 0 offen offset:6" NTS "\n\t"                 \
                 "buffer_load_format_xyz %3, %8, %11, 0 offen offset:8" NTS "\n\t"                 \
                 "buffer_load_format_xyz %4, %9, %10, 0 offen" NTS "\n\t"                          \
                 "buffer_load_format_xyz %5, %9, %10, 0 offen offset:3" NTS "\n\t"                 \
                 "buffer_load_format_xyz %6, %9, %10, 0 offen offset:6" NTS "\n\t"                 \
                 "buffer_load_format_xyz %7, %9, %11, 0 offen offset:8" NTS "\n\t"                 \
                 "s_waitcnt vmcnt(0)"                                                               \
                 : "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]), "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[1][2]), "=&v"(c[1][3]) \
                 : "v"(voff[0]), "v"(voff[1]), "s"(ra), "s"(rb)                                     \
                 : "memory")
            if constexpr (NT && MVFX_STREAM_NT_LOADS) MVFX_TYPED3R(" nt"); else MVFX_TYPED3R("");
#undef MVFX_TYPED3R
#pragma unroll
            for (int u = 0; u < TILE; u++) {
                if (valid[u]) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t T;
                        const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(c[u][j].x, c[u][j].y, c[u][j].z, p, T);
                        w[j] = __builtin_amdgcn_perm(T, 0u, sextant_at(lds.sextant, sel_off));
                    }
                    const u32x3 t = {__builtin_amdgcn_perm(w[1], w[0], 0x04020100u), __builtin_amdgcn_perm(w[2], w[1], 0x05040201u),
                                     __builtin_amdgcn_perm(w[3], w[2], 0x06050402u)};
                    stream_store12<NT>(frame + at[u], t);
                }
            }
        }
    } else {
        const uint32_t tail = width & 3u, total = rows * tail;
        for (uint32_t L = blockIdx.x * (uint32_t)kBlock + threadIdx.x; L < total; L += gridDim.x * (uint32_t)kBlock) {
            const uint32_t row = tail == 1 ? L : (tail == 2 ? L >> 1 : L / 3u), k = L - row * tail;
            const uint32_t at1 = row * stride + gpr * 12u + k * 3u;
            f32x3 c1;
            asm volatile("buffer_load_format_xyz %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(c1) : "v"(at1), "s"(ra) : "memory");
            uint32_t T;
            const uint32_t sel_off = hsvfilter_fast_unit<VARIANT == kFastNeg>(c1.x, c1.y, c1.z, p, T);
            const uint32_t w1 = __builtin_amdgcn_perm(T, 0u, sextant_at(lds.sextant, sel_off)); // [c0 c1 c2 0] in memory order
            uint8_t *q = frame + at1;
            q[0] = (uint8_t)w1;
            q[1] = (uint8_t)(w1 >> 8);
            q[2] = (uint8_t)(w1 >> 16);
        }
    }
}

} // namespace

void launch_hsvfilter3_typed(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                             uint32_t rows, uint64_t stride, const FastConsts &p, uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, bool bgr)
{
#define MVFX_LT3(V, T_, NT_) \
    MVFX_LAUNCH((hsvfilter3_typed_kernel<V, T_, NT_>), grid, dim3(kBlock), 0, stream, fb, width, rows, stride, p, word3a, word3b, frame_bytes, bgr)
#define MVFX_LT3_NT(V, T_) do { if (streaming) MVFX_LT3(V, T_, true); else MVFX_LT3(V, T_, false); } while (0)
    if (tile == 2) { if (neg_shift) MVFX_LT3_NT(kFastNeg, 2); else MVFX_LT3_NT(kFast, 2); }
    else { if (neg_shift) MVFX_LT3_NT(kFastNeg, 1); else MVFX_LT3_NT(kFast, 1); }
#undef MVFX_LT3_NT
#undef MVFX_LT3
}

void launch_hsvfilter3_typed_rows(bool neg_shift, bool streaming, uint32_t n_frames, hipStream_t stream, const FrameBatch &fb, uint32_t width, uint32_t rows,
                                  uint32_t stride, const FastConsts &p, uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, bool bgr)
{
    const uint32_t gpr = width >> 2, magic = gpr > 1 ? (uint32_t)((1ull << 32) / gpr) + 1u : 0u; // floor(2^32 / gpr) + 1: exact quotients while L x gpr < 2^32
    const uint64_t groups = (uint64_t)rows * gpr, tails = (uint64_t)rows * (width & 3u);
    const uint64_t per = (uint64_t)kBlock * 2;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((groups + per - 1) / per, 1), 65535u * 16u);
    (void)tails;
    const dim3 grid(gx, 2, n_frames); // y = 0: whole groups; y = 1: the rows' last pixels (grid-stride: a few workgroups do)
#define MVFX_LR(V, NT_) MVFX_LAUNCH((hsvfilter3_typed_rows_kernel<V, NT_>), grid, dim3(kBlock), 0, stream, fb, width, rows, stride, p, word3a, word3b, frame_bytes, bgr, gpr, magic)
    if (neg_shift) { if (streaming) MVFX_LR(kFastNeg, true); else MVFX_LR(kFastNeg, false); }
    else { if (streaming) MVFX_LR(kFast, true); else MVFX_LR(kFast, false); }
#undef MVFX_LR
}

void launch_hsvfilter_typed(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                            uint32_t rows, uint64_t stride, const FastConsts &p, uint32_t word3, uint32_t frame_bytes, int off, bool bgr)
{
#define MVFX_LT(V, T_, NT_) \
    MVFX_LAUNCH((hsvfilter4_typed_kernel<V, T_, NT_>), grid, dim3(kBlock), 0, stream, fb, width, rows, stride, p, word3, frame_bytes, off, bgr)
#define MVFX_LT_NT(V, T_) do { if (streaming) MVFX_LT(V, T_, true); else MVFX_LT(V, T_, false); } while (0)
    if (tile == 2) { if (neg_shift) MVFX_LT_NT(kFastNeg, 2); else MVFX_LT_NT(kFast, 2); }
    else { if (neg_shift) MVFX_LT_NT(kFastNeg, 1); else MVFX_LT_NT(kFast, 1); }
#undef MVFX_LT_NT
#undef MVFX_LT
}

void launch_hsvfilter_typed_frames(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                                   uint32_t rows, uint64_t stride, const FastConsts &p, const FrameSettingsBatch &fs, uint32_t word3,
                                   uint32_t frame_bytes, int off, bool bgr)
{
#define MVFX_LF(V, T_, NT_) \
    MVFX_LAUNCH((hsvfilter4_typed_frames_kernel<V, T_, NT_>), grid, dim3(kBlock), 0, stream, fb, width, rows, stride, p, fs, word3, frame_bytes, off, bgr)
#define MVFX_LF_NT(V, T_) do { if (streaming) MVFX_LF(V, T_, true); else MVFX_LF(V, T_, false); } while (0)
    if (tile == 2) { if (neg_shift) MVFX_LF_NT(kFastNeg, 2); else MVFX_LF_NT(kFast, 2); }
    else { if (neg_shift) MVFX_LF_NT(kFastNeg, 1); else MVFX_LF_NT(kFast, 1); }
#undef MVFX_LF_NT
#undef MVFX_LF
}

} // namespace mvfx

extern "C" int mvfx_selftest_typed_unorm8(uint32_t *checked_out, uint32_t *mismatches_out)
{
    using namespace mvfx;
    if (!checked_out || !mismatches_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "selftest_typed_unorm8: NULL argument");
    *checked_out = *mismatches_out = 0;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    constexpr uint32_t kLanes = 1024, kBytes = kLanes + 12;
    std::vector<uint8_t> h(kBytes);
    for (uint32_t pos = 0; pos < kBytes; pos++) // position 4 k + r holds (167 k + 59 r) mod 256: all 256 values at every residue r
        h[pos] = (uint8_t)((pos >> 2) * 167u + (pos & 3u) * 59u);
    uint8_t *d = nullptr;
    float *o = nullptr;
    std::vector<float> got((size_t)kLanes * 12);
    MVFX_HIP_TRY(hipMalloc(&d, kBytes));
    if (hipMalloc(&o, got.size() * sizeof(float)) != hipSuccess) { (void)hipFree(d); return fail(MVFX_ERR_OUT_OF_MEMORY, "selftest_typed_unorm8: hipMalloc"); }
    int rc = MVFX_OK;
    uint32_t checked = 0, bad = 0;
    for (int bgr = 0; bgr < 2 && rc == MVFX_OK; bgr++) {
        // the descriptor words of hsvfilter_impl's 3-byte branch (hsv_kernels.hip): A = bytes 0, 1, 2, B = bytes 1, 2, 3 of the four fetched
        const uint32_t r0 = bgr ? 2 : 0, b0 = bgr ? 0 : 2;
        const uint32_t word3a = (4 + r0) | (5u << 3) | ((4 + b0) << 6) | (10u << 15), word3b = (5 + r0) | (6u << 3) | ((5 + b0) << 6) | (10u << 15);
        if (hipMemcpy(d, h.data(), kBytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(o, 0xff, got.size() * sizeof(float)) != hipSuccess) { rc = MVFX_ERR_DEVICE; break; }
        hipLaunchKernelGGL(typed_unorm8_selftest_kernel, dim3(kLanes / kBlock), dim3(kBlock), 0, nullptr, d, o, kBytes, word3a, word3b, kLanes);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(got.data(), o, got.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = MVFX_ERR_DEVICE; break; }
        static const uint32_t first[4] = {0, 3, 6, 9}; // pixel j of a lane starts at byte 3 j (pixel 3: fetched from byte 8, bytes 1..3 delivered)
        for (uint32_t i = 0; i < kLanes; i++)
            for (int j = 0; j < 4; j++)
                for (int ch = 0; ch < 3; ch++) {
                    const uint32_t byte_index = i + first[j] + (ch == 1 ? 1u : (ch == 0 ? r0 : b0));
                    const float want = (float)h[byte_index] / 255.0f; // IEEE division: this file is built with -ffp-contract=off -fno-fast-math
                    const float have = got[(size_t)i * 12 + 3 * j + ch];
                    checked++;
                    if (std::memcmp(&want, &have, sizeof want) != 0) bad++;
                }
    }
    (void)hipFree(d);
    (void)hipFree(o);
    if (rc != MVFX_OK) return fail(rc, "selftest_typed_unorm8: HIP call failed");
    *checked_out = checked;
    *mismatches_out = bad;
    return MVFX_OK;
}
