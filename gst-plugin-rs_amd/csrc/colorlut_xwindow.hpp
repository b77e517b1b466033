// colorlut: the two x-prelerped window kernels as device function bodies (round 3: per-wave windows, round 5: one window per workgroup), shared by
// the HIP kernels of colorlut_window_kernels.hip and the direct-dispatch lane's kernels of direct/colorlut_direct_kernels.hip (round 6) -- same
// arithmetic, same bits; the two differ in how the finished pixels are stored (xwin_store).
#pragma once

#include "colorlut_device.hpp"

namespace mvfx {

typedef const __attribute__((address_space(3))) char *lds_bytes_t;

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
// around a mean colour of the block (18 r bytes since round 5: kXRW): 12 rows of RW x 24 contiguous bytes; pixels outside the window read
// their two entries from the table in global memory.  The per-byte coordinate entries of g and b hold the cell index already
// multiplied by the window's LDS pitch of that axis, so the in-window test and the LDS address are three subtractions, three
// compares, one add3 and one mad.
// (bytes of padding between the y slabs -- a slab is 4 x 432 = 1 728 bytes -- were tried against the bank conflicts of noisy blocks and cost a
// workgroup per CU: profiles/r4, profiles/r5/colorlut_experiments.txt)
// (kXNY = 3 y cells, kXNZ = 3 z cells, kXNZR = 4 z rows -- a pixel reads rows z0 and z0 + 1 -- of a window: colorlut_device.hpp)
// LDS of a wave's window.  The workgroup's total (4 windows + the 4 KB coordinate table) must stay within 32000 bytes: LDS is handed out
// in granules of 1280 bytes and five workgroups per CU need 5 x 25 granules = 160000 <= 163840; one granule more per workgroup costs a
// workgroup per CU (measured: -6 % on every content).  5 184 + 32 spare bytes (the spare de-phases the four waves' windows over the banks).
constexpr uint32_t kXWaveBytes = kXNY * kXPitchY + 32;
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
    constexpr uint32_t kRowP = kXRW * 3 / 2, kPieces = kXNY * kXNZR * kRowP;
#pragma unroll
    for (uint32_t q0 = 0; q0 < kPieces; q0 += 64) {
        const uint32_t q = q0 + lane;
        if (q0 + 64 <= kPieces || q < kPieces) {
            const uint32_t wr = q / kRowP, k = q - wr * kRowP; // window row = dy * (NZ + 1) + dz
            __builtin_amdgcn_global_load_lds((global_void_t)(xtable + (base + ((wr / kXNZR) * (size + 1) + (wr % kXNZR)) * kXRowPieces + k)),
                                             (lds_void_t)(lds_region + q0 * 16), 16, 0, 0);
        }
    }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// volatile: the six 8-byte reads of a pixel stay six ds_read_b64.  Left alone the compiler pairs them into three ds_read2_b64, which the
// LDS serves at HALF the rate (8 array cycles for 16 bytes per lane, 16-lane groups on 32 banks, against 2 x 2 cycles, 32-lane groups on
// 64 banks: MI355X_MICROARCH.md, LDS table)
typedef const volatile __attribute__((address_space(3))) f32x2_t *lds_float2_t;

// How a row of four pixels leaves the kernel.  kStoreNt: a non-temporal store (the HIP launches: the stream's packets end in a release fence that writes
// back whatever the caches hold).  kStoreWtNt: write-through to memory (sc0 sc1) with the non-temporal hint kept -- the direct-dispatch lane's kernels
// (csrc/direct_dispatch.h), whose packets carry no release fence; the caller drains (s_waitcnt vmcnt(0)) before the wave ends.  The two wait states behind
// an inline-asm store of more than 64 bits of data are the ones the compiler inserts behind its own (without them 0.2 % of hsvfilter's pixels came out
// wrong in round 6).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
enum { kStoreNt = 0, kStoreWtNt = 1 };
template <int STORE>
__device__ __forceinline__ void xwin_store(u32x4_t *dst, const u32x4_t &t)
{
    if constexpr (STORE == kStoreWtNt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
    else __builtin_nontemporal_store(t, dst);
}

// The window's anchor is a mean colour of the block, formed out of the pixel registers (no load of its own): every lane offers its pixel (x + 1, y0 + 1),
// a 16 x 4 lattice over the block.  Where the four lanes around the centre agree within kXSpreadLow (sum of absolute byte differences along both
// diagonals) their mean is the anchor; elsewhere the mean of all sixty-four, unless the corner samples differ by more than kXSpread64 along both diagonals
// (an edge: the mean fits neither side, the lane next to the centre stands).  Anchors tried before this one (the centre pixel; four samples by scalar
// loads; by one vector load): profiles/r3/colorlut_anchor4.txt, profiles/r4/colorlut_anchor.txt.
constexpr uint32_t kXSampleRow = 1, kXSpreadLow = 20, kXSpread64 = 120;
// |g - centre g| + |b - centre b| above which an outside pixel counts as far (colorlut_xwg_kernel's test for blocks of uniform-random colours)
constexpr uint32_t kXFar = 64;
// Pixel loads and stores are non-temporal (16 x 4K natural-like 70.7 k -> 73.1 k fps, one frame 21.8 -> 19.1 us: the pixels stream through once,
// the table stays in L2).
template <int STORE>
__device__ __forceinline__ void colorlut_xtile_body(const uint8_t *in, uint8_t *out, uint32_t width, uint32_t height, uint32_t in_stride, uint32_t out_stride,
                                                    const LutParams &p)
{
    constexpr uint32_t RW = kXRW;
    static_assert(RW % 2 == 0 && RW <= 64, "window rows start and end on 16-byte pieces");
    constexpr uint32_t kAcross = 16, kRows = kXRows, kTileW = 64, kTileH = 4 * kRows;
    constexpr uint32_t kWaveBytes = kXWaveBytes;
    __shared__ __attribute__((aligned(16))) uint8_t win[(kBlock / 64) * kWaveBytes];
    __shared__ uint2 coord[512]; // {cell index x LDS pitch, fraction bits} per byte value of the g and b channels
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (giving every XCD a contiguous run of the workgroup order -- whole frames of a batch, a band of a single frame -- so that its L2
    // holds a smaller part of the table: 71.6 k vs 72.3 k fps, one frame 20.8 vs 19.1 us; with non-temporal pixel accesses the table
    // misses are 8 % of the pixel bytes already, FETCH_SIZE 572 MB vs 540 MB per 16 frames)
    const uint32_t gx = blockIdx.x, gy = blockIdx.y;
    const uint32_t bx = (gx * (kBlock / 64) + wave) * kTileW, by = gy * kTileH; // the wave's block
    const uint32_t x = bx + (lane % kAcross) * 4, y0 = by + (lane / kAcross) * kRows;
    const bool whole_block = bx + kTileW <= width && by + kTileH <= height;
    // 1. every pixel of the lane, up front (four 16-byte loads in flight while the window is being fetched)
    uint32_t voff_in = y0 * in_stride + x * 4, voff_out = y0 * out_stride + x * 4; // the lane's byte offsets into rows y0 .. of the frames
    asm volatile("" : "+v"(voff_in), "+v"(voff_out)); // both formed HERE (left alone the compiler re-forms the second one late from a 64-bit x * 4 that it spills)
    uint4 v[kRows];
#pragma unroll
    for (uint32_t row = 0; row < kRows; row++) {
        v[row] = make_uint4(0, 0, 0, 0);
        if (x < width && y0 + row < height) {
            const u32x4_t *src = reinterpret_cast<const u32x4_t *>(in + (size_t)row * in_stride + voff_in); // uniform row base + one 32-bit lane offset
            const u32x4_t t = __builtin_nontemporal_load(src);
            v[row] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
    // 2. the window, anchored at the block's centre pixel (its top-left pixel when the centre lies outside the frame): the pixel and
    // its two coordinate entries come through the scalar cache, so this chain does not wait for the vector loads above
    uint32_t ar, ayp, azp; // anchor: first r byte, y cell x kXPitchY, z row x kXPitchZ
    {
        // (a wave of the last workgroup of a row may lie wholly right of the frame: it reads pixel (0, 0) and stores nothing)
        const uint32_t cxp = bx + kTileW / 2 < width ? bx + kTileW / 2 : bx, cyp = by + kTileH / 2 < height ? by + kTileH / 2 : by;
        const uint32_t coff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bx < width ? cyp * in_stride + cxp * 4 : 0u));
        uint32_t cpx = *reinterpret_cast<const uint32_t *>(in + coff);
        // Samples that cost no memory access at all: every lane's own pixel (x + 1, y0 + 1), out of the registers the pixel loads above
        // fill -- a 16 x 4 lattice over the block.  (A separate sample load fetches lines of its own: with sixteen lanes taking part in it
        // the clean frames lost 4-5 %, profiles/r4/colorlut_anchor.txt.)  The window fill waits for the second row's pixel load.  Where four
        // lanes around the block's centre agree closely (clean content) their mean is the anchor; elsewhere (noise, texture) the mean of all
        // sixty-four -- eight dependent DPP additions on the path the window fill waits for, which clean blocks do not pay.
        if (__builtin_amdgcn_readfirstlane((int)whole_block)) {
            const uint32_t mine = v[kXSampleRow].y;
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 21) & 0xffffffu, i1 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 26) & 0xffffffu,
                           i2 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 37) & 0xffffffu, i3 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 42) & 0xffffffu;
            const uint32_t inner = __builtin_amdgcn_sad_u8(i0, i3, 0u) + __builtin_amdgcn_sad_u8(i1, i2, 0u);
            if (inner <= kXSpreadLow) {
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
            cpx = spread <= kXSpread64 ? mean : ((uint32_t)__builtin_amdgcn_readlane((int)mine, 40) & 0xffffffu);
            }
        }
        cpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cpx);
        const uint32_t cr = cpx & 0xffu;
        // Where the window goes decides how many pixels find their entries in it, never what they compute: the anchor's lattice
        // coordinates may be formed any way.  The two table look-ups of rounds 3 and 4 were scalar loads whose address depends on the
        // pixels -- one more memory round trip on the chain pixel loads -> anchor -> window fill that every wave walks before its first
        // row; the same lattice arithmetic in a handful of VALU operations on the (uniform) anchor colour: +0.7 % calm, +3 % at +-8.
        const float gy = (float)((cpx >> 8) & 0xffu) * (1.0f / 255.0f), bz = (float)((cpx >> 16) & 0xffu) * (1.0f / 255.0f);
        const float ny = fminf(fmaxf(gy * p.scale[1] + p.offset[1], 0.0f), 1.0f) * p.size_m1, nz = fminf(fmaxf(bz * p.scale[2] + p.offset[2], 0.0f), 1.0f) * p.size_m1;
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
    // A row of four pixels goes in two passes of two (round 5: 56-60 VGPRs instead of the 94 of four-pixel passes, which with the 18-byte
    // window puts six workgroups on a CU).  No "far" test here (a block of uniform-random colours, see colorlut_xwg_kernel): pictures like
    // that are the other kernel's -- the content probe decides --, a stray block is served pixel by pixel.
    // (Built and measured in round 4, bit-exact, not shipped: listing the outside pixels per wave (ballot + mbcnt) in the LDS of the dead
    // window and serving them densely from a second 6 x 6 x 6 node window: +8 % at +-8 codes of noise, +7 % at +-16, -3 % at +-5, -7 % on
    // flat bars, and 98 VGPRs -- a wave per SIMD less for every block; profiles/r4/colorlut_dense_pass.txt.)
    auto do_row = [&](const uint32_t row) {
        const bool valid = x < width && y0 + row < height;
        uint32_t px[4] = {v[row].x, v[row].y, v[row].z, v[row].w};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f32x2_t e0[2][3], e1[2][3];
            float ty[2], tz[2];
            bool miss[2];
            bool any_miss = false;
            // the four coordinate reads of a pass before its entry reads (two LDS round trips per pass instead of three: +1 % on calm frames;
            // with four-pixel passes the same idea cost a wave per SIMD and 6 %)
            uint2 egs[2], ebs[2];
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                egs[jj] = coord[(px[2 * h + jj] >> 8) & 0xffu];
                ebs[jj] = coord[256 + ((px[2 * h + jj] >> 16) & 0xffu)];
            }
#pragma unroll
            for (int jj = 0; jj < 2; jj++) asm volatile("" : "+v"(egs[jj]), "+v"(ebs[jj]));
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const uint32_t pxj = px[2 * h + jj];
                const uint2 eg = egs[jj], eb = ebs[jj];
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
            xwin_store<STORE>(dst, t);
        }
    };
#pragma unroll
    for (uint32_t row = 0; row < kRows; row++) do_row(row);
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
static_assert(kWgRW % 2 == 0, "window rows start and end on 16-byte pieces");
static_assert(kWgWinBytes + 4096 + 16 <= 32000, "five workgroups per CU (LDS comes in granules of 1280 bytes: 25 per workgroup)");

template <int STORE>
__device__ __forceinline__ void colorlut_xwg_body(const uint8_t *in, uint8_t *out, uint32_t width, uint32_t height, uint32_t in_stride, uint32_t out_stride,
                                                  const LutParams &p)
{
    constexpr uint32_t kAcross = 16, kRows = kXRows, kTileW = 64, kTileH = 4 * kRows, RW = kWgRW;
    __shared__ __attribute__((aligned(16))) uint8_t win[kWgWinBytes];
    __shared__ uint2 coord[512]; // {cell index x LDS pitch, fraction bits} per byte value of the g and b channels (this kernel's pitches)
    __shared__ uint32_t wave_anchor[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gx = blockIdx.x, gy = blockIdx.y;
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
            const u32x4_t t = __builtin_nontemporal_load(src);
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
        const uint32_t mine = v[kXSampleRow].y;
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
    // (all four pixels of a row in one pass: two passes of two, colorlut_xtile_kernel's form, save registers this kernel's occupancy -- bound by
    // its 27 KB window -- cannot use)
    auto do_row = [&](const uint32_t row) -> bool { // true: row 0 found the block "far" (uniform-random colours) -- nothing served, nothing stored
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
            miss[j] = (dr24 >= RW * 24u) | (dyp >= kWgNY * kWgPitchY) | (dzp >= kWgNZ * kWgPitchZ);
            any_miss = any_miss | miss[j];
            if (row == 0) outside += (uint32_t)__popcll(__ballot(miss[j] & valid));
            const uint32_t off = miss[j] ? 0u : eg.x + eb.x + (r24 + lds_k);
            const lds_float2_t q0 = (lds_float2_t)((lds_bytes_t)&win[0] + off), q1 = (lds_float2_t)((lds_bytes_t)&win[0] + off + kWgPitchZ);
            e0[j][0] = q0[0]; e0[j][1] = q0[1]; e0[j][2] = q0[2];
            e1[j][0] = q1[0]; e1[j][1] = q1[1]; e1[j][2] = q1[2];
        }
        if (row == 0 && outside > 248u) { // wave-uniform, rare
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
                far += (uint32_t)__popcll(__ballot(miss[j] & valid & (__builtin_amdgcn_sad_u8(px[j] & 0x00ffff00u, ccpx & 0x00ffff00u, 0u) > kXFar)));
                alike += (uint32_t)__popcll(__ballot(miss[j] & valid & (((px[j] ^ fpx) & 0x00f0f000u) == 0u)));
            }
            if (far * 4u > outside * 3u && alike * 4u < outside) return true;
        }
        if (any_miss) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
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
        for (int j = 0; j < 4; j++) {
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
        if (valid) {
            u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + (size_t)row * out_stride + voff_out);
            const u32x4_t t = {px[0], px[1], px[2], px[3]};
            xwin_store<STORE>(dst, t);
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
                xwin_store<STORE>(dst, t);
            }
        }
        return;
    }
#pragma unroll
    for (uint32_t row = 1; row < kRows; row++) do_row(row);
}

} // namespace mvfx
