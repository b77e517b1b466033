// colorlut: the LDS window kernels -- RGBA8 (and RGBA64) through a 3-D LUT with the entries a block of pixels needs staged in LDS instead of
// gathered per lane (video/colorlut/src/colorlut/imp.rs:431-449, 493-539: transform_rgba_3d / sample_3d; same arithmetic, same bits as the
// gather kernels of colorlut_kernels.hip), the content probe that chooses between the two x-prelerped forms, and the fused I420 forms.
// Every constant below is the shipped value; the experiments that chose them are recorded in profiles/r3 ... r5 (patches of the variants
// that lost: profiles/r5/*.patch).
#include "colorlut_device.hpp"
#include "colorlut_xwindow.hpp"

namespace mvfx {
namespace {

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
constexpr int kTileCellPitch = 7;
constexpr int kTileWaveLdsFloat4 = kTileNbCells * kTileCellPitch + 2;  // +32 bytes: de-phases the four waves' regions over the banks

// The lattice coordinate of a channel depends on its byte value only: (cell index, fraction) come from a 3 x 256 entry
// table in LDS (built on the host with the same f32 steps, ensure_uploaded) instead of 7 VALU instructions per channel --
// the kernel is VALU-bound once the gathers are gone (rocprofv3: VALUBusy 100 %, profiles/r2/colorlut_tile_counters.txt).
// (Typed buffer loads for u8/255 and several tiles per wave were tried and measured slower here: -4 % and -7 %.  A 5 x 5 x 5
// window for big cubes -- a cell of a 65^3 cube spans only 4 code values, natural-like 4K frame 37.7 us against 24.0 us with
// 33^3 -- costs more than its hits save: 12 KB of cells per tile in twelve wave loads, 48 KB of LDS per workgroup; 65^3
// natural 54.4 us, flat bars 57 us against 30 us, and 33^3 natural 52 us.)

// The 24 floats of the cell (ix, iy, iz): from the wave's LDS window when the cell lies in it, otherwise this lane's own gather
// from the cell table in global memory / L2 (six 16-byte loads).  A quad-cooperative form of the gather (the four lanes of a
// quad fetch one cell with two coalesced loads and hand it over through LDS: 2.8 instead of 6.1 L1 look-ups per pixel) was
// built and measured: no faster on uniform-random colours -- there the L1's miss path is the floor (a cell is two 64-byte L2
// requests, ~0.39 requests per clock per CU) -- and slower when only a few pixels of a tile fall outside
// (profiles/r2/colorlut_random_floor.txt).
// `nbr_base` is an LDS-address-space pointer on purpose: through a generic pointer the six reads become flat loads (the
// kernel then runs at 60 % of its speed).
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

// (the x-prelerped window kernels -- round 3: a window per wave, round 5: one per workgroup -- : colorlut_xwindow.hpp)
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

// the HIP kernels of the two window forms (bodies: colorlut_xwindow.hpp); blockIdx.z = the frame of the batch
// (non-temporal stores, not the write-through ones of csrc/device_store.hpp: measured equal on pictures, 8 % slower on uniform-random colours --
// profiles/r6/store_policy_ab.txt)
__global__ __launch_bounds__(kBlock) void colorlut_xtile_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t height,
                                                                uint32_t in_stride, uint32_t out_stride, LutParams p)
{
    colorlut_xtile_body<kStoreNt>(in_fb.base[blockIdx.z], out_fb.base[blockIdx.z], width, height, in_stride, out_stride, p);
}

__global__ __launch_bounds__(kBlock) void colorlut_xwg_kernel(FrameBatch in_fb, FrameBatch out_fb, uint32_t width, uint32_t height, uint32_t in_stride,
                                                             uint32_t out_stride, LutParams p)
{
    colorlut_xwg_body<kStoreNt>(in_fb.base[blockIdx.z], out_fb.base[blockIdx.z], width, height, in_stride, out_stride, p);
}

// ---------------------------------------------------------------- content probe: which window kernel suits the stream (round 5)
//
// colorlut_xtile_kernel (an 18 x 3 x 3 window per wave, six workgroups per CU) is the faster kernel on calm pictures -- 16 x 4K per launch,
// same box: 80-82 k fps on smooth gradients, 78-79 k with +-3 codes of noise, 69-70 k with +-5 -- and collapses where the colours of a
// 64 x 20 block scatter: 41 k at +-8.  colorlut_xwg_kernel (one 38 x 5 x 5 window per workgroup) runs 74 / 72 / 70.5 / 69 / 47 k at
// +-0 / 3 / 5 / 8 / 16 on the same frames (profiles/r5/colorlut_experiments.txt).  The pictures of a stream resemble their predecessors, so the choice is made from a look at
// an earlier frame: one workgroup, 256 blocks of 64 x 20 pixels spread over the frame, sixteen pixels of each (a 4 x 4 lattice); a block
// is BUSY when the sampled bytes of a channel span more than kProbeSpan codes (sixteen samples of +-3 codes of noise on a gradient span
// about 10, of +-5 about 15, of +-8 about 20).  (colorlut takes RGBA only -- colorlut/imp.rs:122-134 --, so bytes 0..2 of a pixel are its colour.)  More than kProbeBusy busy blocks of 256 make the picture busy.  Every thread of the launch writes nothing
// but thread 0, which stores the verdict into page-locked host memory; the launcher reads that word whenever it launches -- never
// waiting for it -- and runs the probe again every kProbeEvery launches.  Both kernels produce the same bytes: the verdict only moves time.
// Round 6: span 13 / 24 blocks (round 5: 17 / 38, which sent +-5 codes of noise to the per-wave windows although the workgroup window is the faster
// kernel from there on: the driver's sweep read 67.3 k at +-5 below 67.7 k at +-8; tools/exp_colorlut_probe_sweep.py)
constexpr uint32_t kProbeSpan = 13, kProbeBusy = 24; // (kProbeEvery: colorlut_device.hpp)

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
// colorlut_xtile_kernel -- window of kXRW r bytes x 3 y cells x 4 z rows around the first pixel of lane 36 (lane 0's when
// the block's centre lies outside the frame), filled by global_load_lds, premultiplied coordinate entries for g and b.
__global__ __launch_bounds__(kI420Block) void colorlut_i420_xtile_kernel(I420Planes pl, uint32_t width, uint32_t height, LutParams p,
                                                                         YuvToRgbCoef kin, RgbToYuvCoef kout)
{
    constexpr uint32_t RW = kXRW, kWaveBytes = kXWaveBytes;
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
    // a block that lies wholly inside the frame is anchored at the MEAN of its 64 lanes' first pixels (an 8 x 8 lattice over the
    // 64 x 16 block, already in registers) unless its corners say an edge runs through it: colorlut_xtile_kernel's anchor
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
        if (__builtin_amdgcn_sad_u8(q0, q3, 0u) + __builtin_amdgcn_sad_u8(q1, q2, 0u) <= kXSpread64) fpx = mean;
    }
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
} // namespace

void launch_colorlut_xtable_build(const float4 *cube, const uint32_t *tile_tables, uint32_t size, float *xtable, uint64_t entries)
{
    // (not MVFX_LAUNCH: part of the LUT's upload, not of a frame's work -- the frame's completion event stays out of it)
    hipLaunchKernelGGL(colorlut_xtable_build_kernel, dim3((uint32_t)((entries + 255) / 256)), dim3(256), 0, nullptr, cube, tile_tables, size, xtable);
}

void launch_colorlut_xtile(dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width, uint32_t height, uint32_t in_stride,
                           uint32_t out_stride, const LutParams &p)
{
    MVFX_LAUNCH(colorlut_xtile_kernel, grid, dim3(kBlock), 0, st, in, out, width, height, in_stride, out_stride, p);
}

void launch_colorlut_xwg(dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width, uint32_t height, uint32_t in_stride,
                         uint32_t out_stride, const LutParams &p)
{
    MVFX_LAUNCH(colorlut_xwg_kernel, grid, dim3(kBlock), 0, st, in, out, width, height, in_stride, out_stride, p);
}

void launch_colorlut_probe(hipStream_t st, const uint8_t *frame, uint32_t width, uint32_t height, uint32_t stride, uint32_t *verdict)
{
    hipLaunchKernelGGL(colorlut_probe_kernel, dim3(1), dim3(256), 0, st, frame, width, height, stride, verdict); // (not MVFX_LAUNCH: no part of the frame's work)
}

void launch_colorlut_tile(bool wide, bool le, bool narrow, dim3 grid, hipStream_t st, const FrameBatch &in, const FrameBatch &out, uint32_t width,
                          uint32_t height, uint32_t in_stride, uint32_t out_stride, const LutParams &p)
{
#define MVFX_TK(WIDE, LE, A, R) MVFX_LAUNCH((colorlut_tile_kernel<WIDE, LE, A, R>), grid, dim3(kBlock), 0, st, in, out, width, height, in_stride, out_stride, p)
    if (wide) { if (le) MVFX_TK(true, true, 16, 4); else MVFX_TK(true, false, 16, 4); }
    else if (narrow) MVFX_TK(false, true, 8, 2);
    else MVFX_TK(false, true, 16, 4);
#undef MVFX_TK
}

void launch_colorlut_i420_window(bool xtile, dim3 grid, hipStream_t st, const I420Planes &pl, uint32_t width, uint32_t height, const LutParams &p,
                                 const YuvToRgbCoef &kin, const RgbToYuvCoef &kout)
{
    if (xtile) MVFX_LAUNCH(colorlut_i420_xtile_kernel, grid, dim3(kI420Block), 0, st, pl, width, height, p, kin, kout);
    else MVFX_LAUNCH(colorlut_i420_tile_kernel, grid, dim3(kI420Block), 0, st, pl, width, height, p, kin, kout);
}

} // namespace mvfx
