// f32 pipeline of the SSIM-family distance behind videocompare's `hash-algo=dssim` (round 3).
//
// dssim-core (the crate the reference calls, videocompare/hashed_image.rs:49-59,72-75) is an f32 library; round 2's
// device path carried f64 planes through five kernel types: 8.7 GB of HBM traffic per 8K pair against 265 MB of compulsory
// input (profiles/r3/traffic_summary.txt), 1.83 ms per pair.  This file keeps everything per pixel in f32 and in LDS /
// registers, f64 only in the reductions:
//
//   one kernel per pyramid level (ssim32_level_kernel): a workgroup owns a 60 x 28 tile, fetches the 64 x 32 haloed source
//   pixels of BOTH images (level 0: the frame bytes; level >= 1: the previous level's linear-RGB f32 planes), converts them
//   to the Lab-like planes the metric runs on -- as (image A, image B) pairs, so the arithmetic is packed f32 on aligned
//   VGPR pairs and the pairs go to LDS with one 8-byte store -- writes the 2x2 box average of the linear values as the
//   next level's planes (the pyramid comes out of the same pass), then runs the separable 5x5 binomial window as a sliding
//   window down each column: per row five 8-byte LDS reads and the horizontal sums of (s, d) and (s^2, d^2) -- s, d the sum and
//   difference of the two images' values -- in registers, two packed operations each, a ring of five rows for the vertical sums; the SSIM terms of the three channels are summed in
//   registers and the map value written once (f32).  No Lab plane, no window sum ever goes to memory.
//   HBM traffic per 8K pair: 2 x 133 MB of bytes (x1.2 halo re-reads, mostly L2 hits) + 2 x 100 MB of level-1 planes written
//   and read + the f32 maps (133 + 33 + ... MB written, read once by the deviation pass).
//
//   Cancellation: var = E[x^2] - E[x]^2 loses digits in f32 when the window is flat and bright; every tile therefore
//   subtracts a per-tile, per-channel constant (image A's value at the tile centre) before squaring -- the variance and
//   covariance are invariant, the means get it added back.
//
//   Near-identical frames: 1 - ssim is tiny, and an f32 quotient next to 1.0 cannot hold it (6e-8 absolute).  The kernel therefore
//   computes the DEFICIT of every term directly.  With a = (m1 - m2)^2 and b = Var(x1 - x2) over the window (LDS holds the sum
//   and the difference of the two images' values, the window sums are those of s, d, s^2, d^2 -- all packed):  2 m1 m2 + C1 = (m1^2 + m2^2 + C1) - a  and  2 s12 + C2 = (s11 + s22 + C2) - b, so
//   1 - term = (a sd + b (ld - a)) / (ld sd)  with ld, sd the two denominators -- every factor is formed from small quantities
//   without cancellation, the deficit has f32 RELATIVE accuracy, and the map in memory is the deficit map (mean and mean absolute
//   deviation are the same numbers either way).  Identical frames: x1 - x2 = 0 everywhere, a = b = 0, deficit exactly 0,
//   distance exactly 0 (tests/videocompare.rs:141-182).
//
// PARITY UNPINNED against the crate: the checker is oracle/ssim_oracle.c (f64); the f32 device value agrees with
// it to ~1e-6 relative (tests/test_ssim_gpu.py states the tolerance).  No FMA contraction in this file: the fused operations are
// written out.
#include "mvfx_internal.h"
#include "ssim32.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace mvfx {
namespace ssim32 {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kTW = 60, kTH = 28;          // interior of a tile (outputs)
constexpr int kRW = kTW + 4, kRH = kTH + 4; // haloed tile: 64 x 32
constexpr int kThreads = 256;
constexpr int kSegRows = 7, kSegs = kTH / kSegRows; // blur tasks: (column, segment of 7 rows), 240 of the 256 threads
constexpr int kBlocksX = kRW / 2, kBlocksY = kRH / 2; // 2x2 pixel blocks of the haloed tile: 32 x 16 = 512, two per thread
static_assert(kSegs * kSegRows == kTH && kTW * kSegs <= kThreads && kBlocksX * kBlocksY == 2 * kThreads, "tile shape");
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// cbrt(t) for t in (216/24389, ~1.1]: exp2(log2(t) / 3) from the hardware transcendentals (relative error ~3e-7), one Newton step
// r <- r - (r^3 - t) / (3 r^2) written division-free with y = t^(-1/3): y' = y (4 - t y^3) / 3, cbrt = t y'^2
__device__ __forceinline__ float cbrt_unit(float t)
{
    float y = __builtin_amdgcn_exp2f(-0.33333334f * __builtin_amdgcn_logf(t)); // t^(-1/3)
    const float y3 = y * y * y;
    y = y * __builtin_fmaf(-0.33333334f * t, y3, 1.3333334f);
    return t * (y * y);
}

__device__ __forceinline__ float lab_f(float t)
{
    const float eps = 216.0f / 24389.0f, kappa = 24389.0f / 27.0f;
    const float lin = __builtin_fmaf(kappa / 116.0f, t, 16.0f / 116.0f); // (kappa t + 16) / 116 in one operation
    // both sides are always computed and the result is a select: as a branch (what the compiler makes of `t > eps ? cbrt : lin`
    // when it may sink the transcendentals) every one of the 48 cube roots of a lane became its own basic block with a serial
    // log -> exp -> Newton chain and nothing to overlap it with (round 3: 470 -> ... us for the 8K level-0 launch)
    // (no guard on t: for t <= eps -- including 0, where the logarithm is -inf and the Newton step makes a NaN -- the select below
    // takes `lin` and never looks at the other operand)
    float cb = cbrt_unit(t);
    asm volatile("" : "+v"(cb));
    return t > eps ? cb : lin;
}

#ifndef MVFX_SSIM_READ2
#define MVFX_SSIM_READ2 0 // 1: let the compiler pair the window's 8-byte LDS reads into ds_read2_b64 (round 3)
#endif
#ifndef MVFX_SSIM_CBRT_TABLE
#define MVFX_SSIM_CBRT_TABLE 0 // 1: cube roots from an LDS seed table + two Newton steps instead of log2 / exp2 + one step.  Measured: SLOWER,
                               // 8K pair 0.660 -> 0.739 ms (1515 -> 1353 pairs/s): six more random LDS reads per pixel pair and a longer
                               // dependent chain cost more than the two quarter-rate instructions they replace (profiles/r3/ssim32_cbrt_table_ab.txt)
#endif
#if MVFX_SSIM_CBRT_TABLE
// The seed table: t^(-1/3) at the centre of every bin of the top 15 bits of a float (sign, exponent, 6 mantissa bits) between
// eps = 216/24389 and 1.1 -- 447 bins, relative width 1/64, seed error <= 0.27 %.  Two Newton steps y <- y (4 - t y^3) / 3 square that
// to 1.5e-5 and then below the rounding of an f32; log2 / exp2 (quarter rate) are gone from the pixel path.
constexpr uint32_t kCbrtLo = 0x3C111A6Cu >> 17; // bits(216 / 24389 = 0.008856452) >> 17
constexpr uint32_t kCbrtBins = 448;

__device__ __forceinline__ void cbrt_table_fill(float *table, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t i = tid; i < kCbrtBins; i += nthreads) {
        const float centre = __uint_as_float(((kCbrtLo + i) << 17) | (1u << 16));
        table[i] = __builtin_amdgcn_exp2f(-0.33333334f * __builtin_amdgcn_logf(centre));
    }
}

__device__ __forceinline__ float cbrt_seeded(float t, const float *table)
{
    const uint32_t bin = min((__float_as_uint(t) >> 17) - kCbrtLo, kCbrtBins - 1); // below eps: wraps, clamped; the caller's select drops it
    float y = table[bin];
    const float a = -0.33333334f * t;
    y = y * __builtin_fmaf(a, y * y * y, 1.3333334f);
    y = y * __builtin_fmaf(a, y * y * y, 1.3333334f);
    return t * (y * y);
}

#endif

__device__ __forceinline__ float lab_f(float t, const float *cbrt_table)
{
#if MVFX_SSIM_CBRT_TABLE
    const float eps = 216.0f / 24389.0f, kappa = 24389.0f / 27.0f;
    const float lin = __builtin_fmaf(kappa / 116.0f, t, 16.0f / 116.0f);
    float cb = cbrt_seeded(t, cbrt_table);
    asm volatile("" : "+v"(cb));
    return t > eps ? cb : lin;
#else
    return lab_f(t);
#endif
}

// linear RGB (premultiplied) of the same pixel of both images -> the three planes of both, as pairs (oracle: to_lab)
__device__ __forceinline__ void to_lab2(f2 r, f2 g, f2 b, f2 out[3], const float *ct)
{
    // the white-point divisions are folded into the matrix rows (X / 0.9505, Z / 1.089): one rounding less per value than the oracle's
    // two steps, inside the tolerance of this pipeline (tests/test_ssim_gpu.py)
    const f2 X = fma2(splat(0.1805f / 0.9505f), b, fma2(splat(0.3576f / 0.9505f), g, splat(0.4124f / 0.9505f) * r));
    const f2 Y = fma2(splat(0.0722f), b, fma2(splat(0.7152f), g, splat(0.2126f) * r));
    const f2 Z = fma2(splat(0.9505f / 1.089f), b, fma2(splat(0.1192f / 1.089f), g, splat(0.0193f / 1.089f) * r));
    const f2 fx = {lab_f(X.x, ct), lab_f(X.y, ct)}, fy = {lab_f(Y.x, ct), lab_f(Y.y, ct)}, fz = {lab_f(Z.x, ct), lab_f(Z.y, ct)};
    out[0] = fma2(splat(1.16f), fy, splat(-0.16f));
    out[1] = fma2(splat(500.0f / 220.0f), fx - fy, splat(86.2f / 220.0f));
    out[2] = fma2(splat(200.0f / 220.0f), fy - fz, splat(107.9f / 220.0f));
}

struct LevelArgs {
    // source of both images: MODE 0 / 1 frame bytes, MODE 2 the three linear planes of this level (pitch = w)
    const uint8_t *bytes[2];
    uint64_t stride[2];
    const float *lin[2][3];
    const float *lut;       // sRGB byte -> linear f32 (256 entries)
    int w, h;               // size of this level
    int cover_lo;           // first row of the tile grid (even)
    int rd_lo, rd_hi;       // rows of the source that exist for this launch (a band: the rows the previous level produced)
    int y0, y1;             // rows of the map this launch produces (the band)
    float *map;             // w x h, absolutely indexed
    double *sum;            // kSlots accumulators of the map sum
    float *nxt[2][3];       // next level's linear planes ((w/2) x (h/2)), NULL at the last level
    int nw, nh, ny_lo, ny_hi; // next level's size and the rows of it this launch must produce
    int vec8;               // two horizontally adjacent source pixels (x even) may be fetched with one 8-byte load
};

// One pixel pair -> linear premultiplied RGB of image A (x lanes) and image B (y lanes).
template <int MODE, int BPP>
__device__ __forceinline__ void load_linear(const LevelArgs &A, const float *s_lut, int x, int y, f2 &r, f2 &g, f2 &b)
{
    if (MODE == 2) {
        const size_t i = (size_t)y * A.w + x;
        r = (f2){A.lin[0][0][i], A.lin[1][0][i]};
        g = (f2){A.lin[0][1][i], A.lin[1][1][i]};
        b = (f2){A.lin[0][2][i], A.lin[1][2][i]};
        return;
    }
    uint32_t c[2][4];
#pragma unroll
    for (int im = 0; im < 2; im++) {
        const uint8_t *p = A.bytes[im] + (uint64_t)y * A.stride[im] + (uint64_t)x * BPP;
        if (MODE == 0) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
            c[im][0] = v & 0xffu; c[im][1] = (v >> 8) & 0xffu; c[im][2] = (v >> 16) & 0xffu; c[im][3] = v >> 24;
        } else {
            c[im][0] = p[0]; c[im][1] = p[1]; c[im][2] = p[2];
            c[im][3] = BPP == 4 ? p[3] : 255u;
        }
    }
    const f2 a = (f2){(float)c[0][3], (float)c[1][3]} * splat(1.0f / 255.0f); // 255 * (1/255) rounds to exactly 1
    r = (f2){s_lut[c[0][0]], s_lut[c[1][0]]} * a;
    g = (f2){s_lut[c[0][1]], s_lut[c[1][1]]} * a;
    b = (f2){s_lut[c[0][2]], s_lut[c[1][2]]} * a;
}

// The two pixels (x, y), (x + 1, y) of a 2x2 block's row (x even, may lie outside the frame: edge replication) of both images.
template <int MODE, int BPP>
__device__ __forceinline__ void load_linear_pair(const LevelArgs &A, const float *s_lut, int x, int y, f2 r[2], f2 g[2], f2 b[2])
{
    if (A.vec8 && x >= 0 && x + 1 < A.w) {
        if (MODE == 2) {
            const size_t i = (size_t)y * A.w + x;
            f2 v[2][3];
#pragma unroll
            for (int im = 0; im < 2; im++)
#pragma unroll
                for (int c = 0; c < 3; c++) v[im][c] = *reinterpret_cast<const f2 *>(A.lin[im][c] + i);
            r[0] = (f2){v[0][0].x, v[1][0].x}; r[1] = (f2){v[0][0].y, v[1][0].y};
            g[0] = (f2){v[0][1].x, v[1][1].x}; g[1] = (f2){v[0][1].y, v[1][1].y};
            b[0] = (f2){v[0][2].x, v[1][2].x}; b[1] = (f2){v[0][2].y, v[1][2].y};
            return;
        }
        if (MODE == 0) {
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            u2 px[2];
#pragma unroll
            for (int im = 0; im < 2; im++)
                px[im] = *reinterpret_cast<const u2 *>(A.bytes[im] + (uint64_t)y * A.stride[im] + (uint64_t)x * 4);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t va = k ? px[0].y : px[0].x, vb = k ? px[1].y : px[1].x;
                const f2 al = (f2){(float)(va >> 24), (float)(vb >> 24)} * splat(1.0f / 255.0f);
                r[k] = (f2){s_lut[va & 0xffu], s_lut[vb & 0xffu]} * al;
                g[k] = (f2){s_lut[(va >> 8) & 0xffu], s_lut[(vb >> 8) & 0xffu]} * al;
                b[k] = (f2){s_lut[(va >> 16) & 0xffu], s_lut[(vb >> 16) & 0xffu]} * al;
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
        load_linear<MODE, BPP>(A, s_lut, min(max(x + k, 0), A.w - 1), y, r[k], g[k], b[k]);
}

// weighted sum with the binomial row (1, 4, 6, 4, 1); the 1/16 per direction is applied once at the end (1/256, exact)
__device__ __forceinline__ f2 binom5(f2 a, f2 b, f2 c, f2 d, f2 e)
{
    return fma2(splat(6.0f), c, fma2(splat(4.0f), b + d, a + e));
}
// n / d to ~1 ulp for the operands that occur here (finite, d > 0): reciprocal, one residual correction
__device__ __forceinline__ float quotient(float n, float d)
{
    const float rc = __builtin_amdgcn_rcpf(d);
    const float q = n * rc;
    const float rem = __builtin_fmaf(-q, d, n);
    return __builtin_fmaf(rem, rc, q);
}

template <int MODE, int BPP>
__global__ __launch_bounds__(kThreads) void ssim32_level_kernel(LevelArgs A)
{
    __shared__ __attribute__((aligned(16))) f2 raw[3][kRH][kRW]; // 48 KiB: per pixel and plane (x1' + x2', x1 - x2), x' = x - centre
    __shared__ float s_lut[256];
    __shared__ f2 s_centre[3];
    __shared__ double s_part[kThreads / 64];
    const int tx0 = blockIdx.x * kTW, ty0 = A.cover_lo + blockIdx.y * kTH;
    __shared__ float s_cbrt[MVFX_SSIM_CBRT_TABLE ? 448 : 1];
#if MVFX_SSIM_CBRT_TABLE
    cbrt_table_fill(s_cbrt, threadIdx.x, kThreads);
#endif
    if (MODE != 2) s_lut[threadIdx.x] = A.lut[threadIdx.x];
    if (MVFX_SSIM_CBRT_TABLE || MODE != 2) __syncthreads();

    // ---- fetch + convert: thread t owns the 2x2 blocks t and t + 256 of the haloed tile ---------------------------------
    f2 lab[2][4][3]; // [block][pixel of the block][channel]
#pragma unroll
    for (int k2 = 0; k2 < 2; k2++) {
        const int blk = threadIdx.x + k2 * kThreads, bx = blk % kBlocksX, by = blk / kBlocksX;
        f2 box[3] = {splat(0.0f), splat(0.0f), splat(0.0f)};
#pragma unroll
        for (int j = 0; j < 2; j++) { // the two rows of the block
            // edge replication (oracle: clampi); rows are clamped to the rows that exist for this launch, which is the frame for a
            // whole-frame call and differs from it only in rows whose outputs are discarded for a band
            const int x = tx0 - 2 + 2 * bx, y = min(max(ty0 - 2 + 2 * by + j, A.rd_lo), A.rd_hi - 1);
            f2 r[2], g[2], b[2];
            load_linear_pair<MODE, BPP>(A, s_lut, x, y, r, g, b);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                box[0] += r[k]; box[1] += g[k]; box[2] += b[k];
                to_lab2(r[k], g[k], b[k], lab[k2][2 * j + k], s_cbrt);
            }
        }
        // the pyramid: 2x2 box of the LINEAR values of the interior blocks -> next level's planes (oracle: downsample)
        if (A.nxt[0][0] != nullptr && bx >= 1 && bx <= kTW / 2 && by >= 1 && by <= kTH / 2) {
            const int X = tx0 / 2 - 1 + bx, Y = (ty0 - 2) / 2 + by;
            if (X < A.nw && Y >= A.ny_lo && Y < A.ny_hi) {
                const size_t o = (size_t)Y * A.nw + X;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const f2 v = box[c] * splat(0.25f);
                    A.nxt[0][c][o] = v.x;
                    A.nxt[1][c][o] = v.y;
                }
            }
        }
    }
    // the tile's centring constants: image A's planes at the haloed tile's centre pixel (block (16, 8), its first pixel)
    if (threadIdx.x == (kBlocksY / 2) * kBlocksX + kBlocksX / 2 - kThreads) {
#pragma unroll
        for (int c = 0; c < 3; c++) s_centre[c] = splat(lab[1][0][c].x);
    }
    __syncthreads();
    f2 centre[3];
#pragma unroll
    for (int c = 0; c < 3; c++) centre[c] = s_centre[c];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int blk = threadIdx.x + k * kThreads, bx = blk % kBlocksX, by = blk / kBlocksX;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // LDS holds, per pixel, the pair (s, d) = (x1' + x2', x1 - x2) of the centred values: everything the SSIM term needs is a
            // window sum of s, d, s^2 or d^2 (see below), so the whole window arithmetic is packed f32 on these pairs.
            // two pixels of a row = 16 contiguous bytes of LDS
            typedef float f4 __attribute__((ext_vector_type(4)));
            f2 q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f2 v = lab[k][j][c], vc = v - centre[c];
                q[j] = (f2){vc.x + vc.y, v.x - v.y};
            }
            *reinterpret_cast<f4 *>(&raw[c][2 * by][2 * bx]) = (f4){q[0].x, q[0].y, q[1].x, q[1].y};
            *reinterpret_cast<f4 *>(&raw[c][2 * by + 1][2 * bx]) = (f4){q[2].x, q[2].y, q[3].x, q[3].y};
        }
    }
    __syncthreads();

    // ---- separable 5x5 binomial window, sliding down the column; SSIM term of the three channels ----------------------------
    float acc[kSegRows];
#pragma unroll
    for (int r = 0; r < kSegRows; r++) acc[r] = 0.0f;
    const int col = threadIdx.x % kTW, seg = threadIdx.x / kTW;
    if (seg < kSegs) {
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            // With s = x1' + x2' and d = x1 - x2 (x' = x - centre):   m1 - m2 = E[d],   m1' + m2' = E[s],
            //   m1^2 + m2^2 = ((m1 + m2)^2 + (m1 - m2)^2) / 2,   E[x1'^2] + E[x2'^2] = (E[s^2] + E[d^2]) / 2,
            //   s11 + s22 = E[x1'^2] + E[x2'^2] - (m1'^2 + m2'^2),   Var(x1 - x2) = E[d^2] - E[d]^2
            // -- the term never needs s11, s22 or the means separately.  Window sums carry the weight 256 un-normalised.
            f2 hp[5], hq[5]; // ring of the horizontal sums of (s, d) and (s^2, d^2)
            const float c512 = 512.0f * centre[c].x; // 256 (m1 + m2) = MS + 512 centre
#pragma unroll
            for (int j = 0; j < kSegRows + 4; j++) {
                // volatile: five ds_read_b64 (consecutive lanes read consecutive 8-byte pairs: conflict-free, 2 LDS cycles each).  Left alone
                // the compiler pairs them into ds_read2_b64, which the LDS serves at half the rate (MI355X_MICROARCH.md, LDS table)
#if MVFX_SSIM_READ2
                typedef const __attribute__((address_space(3))) f2 *lds_f2_t;
#else
                typedef const volatile __attribute__((address_space(3))) f2 *lds_f2_t; // (a generic volatile pointer becomes flat loads)
#endif
                const lds_f2_t row = (lds_f2_t)&raw[c][seg * kSegRows + j][col];
                const f2 p0 = row[0], p1 = row[1], p2 = row[2], p3 = row[3], p4 = row[4];
                hp[j % 5] = binom5(p0, p1, p2, p3, p4);
                hq[j % 5] = binom5(p0 * p0, p1 * p1, p2 * p2, p3 * p3, p4 * p4);
                if (j >= 4) {
                    // window rows j-4 .. j (ring order is irrelevant to the symmetric weights except for the centre: row j-2)
                    const int ra = (j - 4) % 5, rb = (j - 3) % 5, rm = (j - 2) % 5, rd = (j - 1) % 5, re = j % 5;
                    const f2 M = binom5(hp[ra], hp[rb], hp[rm], hp[rd], hp[re]);   // 256 (m1' + m2', m1 - m2)
                    const f2 Q = binom5(hq[ra], hq[rb], hq[rm], hq[rd], hq[re]);   // 256 (E[s^2], E[d^2])
                    const f2 M2 = M * M;
                    const float a = M2.y;                                          // 65536 (m1 - m2)^2
                    const float b = __builtin_fmaf(256.0f, Q.y, -a);               // 65536 Var(x1 - x2)
                    const float var = __builtin_fmaf(128.0f, Q.x + Q.y, -0.5f * (M2.x + M2.y)); // 65536 (s11 + s22)
                    const float ms = M.x + c512;                                   // 256 (m1 + m2)
                    const float ld = __builtin_fmaf(0.5f, __builtin_fmaf(ms, ms, a), 65536.0f * kC1); // 65536 (m1^2 + m2^2 + C1)
                    const float sd = var + 65536.0f * kC2;
                    // 1 - (ld - a)(sd - b) / (ld sd) = (a sd + b (ld - a)) / (ld sd)
                    acc[j - 4] += quotient(__builtin_fmaf(a, sd, b * (ld - a)), ld * sd);
                }
            }
        }
    }
    double total = 0.0;
    if (seg < kSegs) {
        const int x = tx0 + col;
        float part = 0.0f;
#pragma unroll
        for (int r = 0; r < kSegRows; r++) {
            const int y = ty0 + seg * kSegRows + r;
            if (x < A.w && y >= A.y0 && y < A.y1) {
                const float val = acc[r] * (1.0f / 3.0f); // the DEFICIT 1 - ssim of this pixel (mean of the three channels)
                A.map[(size_t)y * A.w + x] = val;
                part += val; // <= 7 values
            }
        }
        total = (double)part;
    }
    for (int off = 32; off > 0; off >>= 1)
        total += __shfl_down(total, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < kThreads / 64; i++) t += s_part[i];
        atomicAdd(A.sum + ((blockIdx.x + blockIdx.y * gridDim.x) % kSlots), t);
    }
}

// second pass: sum of |map - mean| over the band = sum of |deficit - (1 - mean)| (the deficits are f32; difference and sum f64);
// all levels in one launch (blockIdx.y = level): five launches of 5...40 us of work each were launch-bound
struct DevArgs {
    const float *map[kScales];
    size_t first[kScales], count[kScales]; // the band's elements of every level's map
    double avg_deficit[kScales];
    double *sum; // [kScales][kSlots]
};

__global__ __launch_bounds__(kThreads) void ssim32_dev_kernel(DevArgs D)
{
    __shared__ double s_part[kThreads / 64];
    const int s = blockIdx.y;
    const float *map = D.map[s] + D.first[s];
    const size_t n = D.count[s];
    const double avg = D.avg_deficit[s];
    double t = 0.0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
        t += fabs((double)map[i] - avg);
    for (int off = 32; off > 0; off >>= 1)
        t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0 && n) {
        double sum = 0.0;
        for (int i = 0; i < kThreads / 64; i++) sum += s_part[i];
        atomicAdd(D.sum + (size_t)s * kSlots + (blockIdx.x % kSlots), sum);
    }
}

// Per-thread scratch, kept while the frame size stays the same: linear planes of levels 1..4 (both images), the five maps.
struct State {
    std::vector<void *> allocations;
    int w0 = 0, h0 = 0, device = -1;
    float *lin[kScales][2][3] = {};
    float *map[kScales] = {};
    int w[kScales] = {}, h[kScales] = {}, y0[kScales] = {}, y1[kScales] = {};
    int scales = 0;
    double *d_sums = nullptr; // [2 passes][kScales][kSlots]
    float *d_lut = nullptr;
    void release()
    {
        for (void *p : allocations) (void)hipFree(p);
        allocations.clear();
        scales = 0;
        w0 = h0 = 0;
    }
    ~State() { release(); }
};
thread_local State t_state;

int dalloc(State &S, size_t bytes, void **out)
{
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "ssim: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    S.allocations.push_back(*out);
    return MVFX_OK;
}

int ensure_scratch(State &S, int w0, int h0, hipStream_t st)
{
    int dev = 0;
    MVFX_HIP_TRY(hipGetDevice(&dev));
    if (S.w0 == w0 && S.h0 == h0 && S.device == dev && !S.allocations.empty())
        return MVFX_OK;
    S.release();
    S.device = dev;
    if (int rc = dalloc(S, sizeof(double) * 2 * kScales * kSlots, reinterpret_cast<void **>(&S.d_sums)); rc != MVFX_OK) return rc;
    if (int rc = dalloc(S, sizeof(float) * 256, reinterpret_cast<void **>(&S.d_lut)); rc != MVFX_OK) return rc;
    float lut[256];
    for (int i = 0; i < 256; i++) { // oracle: srgb_to_linear, rounded once to f32
        const double x = i / 255.0;
        lut[i] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
    }
    MVFX_HIP_TRY(hipMemcpyAsync(S.d_lut, lut, sizeof(lut), hipMemcpyHostToDevice, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st)); // `lut` is a stack buffer
    int w = w0, h = h0;
    for (int s = 0; s < kScales; s++) {
        if (s > 0) {
            if (w / 2 < 8 || h / 2 < 8) break;
            w /= 2; h /= 2;
            for (int i = 0; i < 2; i++)
                for (int c = 0; c < 3; c++)
                    if (int rc = dalloc(S, sizeof(float) * (size_t)w * h, reinterpret_cast<void **>(&S.lin[s][i][c])); rc != MVFX_OK) return rc;
        }
        if (int rc = dalloc(S, sizeof(float) * (size_t)w * h, reinterpret_cast<void **>(&S.map[s])); rc != MVFX_OK) return rc;
    }
    S.w0 = w0; S.h0 = h0;
    return MVFX_OK;
}

template <int MODE, int BPP>
void launch_level(const LevelArgs &A, int cover_hi, hipStream_t st)
{
    const dim3 grid((A.w + kTW - 1) / kTW, (cover_hi - A.cover_lo + kTH - 1) / kTH);
    MVFX_LAUNCH((ssim32_level_kernel<MODE, BPP>), grid, dim3(kThreads), 0, st, A);
}

int read_slots(const double *d_slots, double out[kScales], hipStream_t st)
{
    // into page-locked memory (one block per thread, kept): a D2H copy to pageable memory is staged and blocks twice
    static thread_local double *pinned = nullptr;
    if (!pinned) {
        void *q = nullptr;
        if (hipHostMalloc(&q, sizeof(double) * kScales * kSlots, hipHostMallocDefault) == hipSuccess) pinned = static_cast<double *>(q);
        else (void)hipGetLastError();
    }
    std::vector<double> pageable;
    double *slots = pinned;
    if (!slots) {
        pageable.resize((size_t)kScales * kSlots);
        slots = pageable.data();
    }
    MVFX_HIP_TRY(hipMemcpyAsync(slots, d_slots, sizeof(double) * kScales * kSlots, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    for (int s = 0; s < kScales; s++) {
        out[s] = 0.0;
        for (int k = 0; k < kSlots; k++) out[s] += slots[(size_t)s * kSlots + k];
    }
    return MVFX_OK;
}

} // namespace

int partial_sums(const mvfx_frame *const fr[2], uint32_t row_begin, uint32_t row_end, double sums_out[5], double counts_out[5],
                 uint32_t *n_scales_out, hipStream_t st)
{
    const int w0 = (int)fr[0]->width, h0 = (int)fr[0]->height;
    State &S = t_state;
    if (int rc = ensure_scratch(S, w0, h0, st); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemsetAsync(S.d_sums, 0, sizeof(double) * 2 * kScales * kSlots, st));

    // geometry: size of every level, the band's map rows [y0, y1) there, and the rows [lo, hi) its tile grid has to cover: the
    // band itself and twice what the next coarser level READS (its cover +- 2 rows of window halo)
    int n_scales = 0, lo[kScales], hi[kScales];
    for (int s = 0, w = w0, h = h0; s < kScales; s++) {
        if (s > 0) {
            if (w / 2 < 8 || h / 2 < 8) break;
            w /= 2; h /= 2;
        }
        S.w[s] = w; S.h[s] = h;
        S.y0[s] = std::min((int)(row_begin >> s), h);
        S.y1[s] = row_end == (uint32_t)h0 ? h : std::min((int)(row_end >> s), h);
        n_scales = s + 1;
    }
    int read_lo = 0, read_hi = 0; // rows of level s + 1 that level s + 1 reads (empty at the start)
    for (int s = n_scales - 1; s >= 0; s--) {
        lo[s] = S.y0[s]; hi[s] = S.y1[s];
        if (hi[s] <= lo[s]) { lo[s] = 0; hi[s] = 0; }
        if (read_hi > read_lo) {
            const int need_lo = 2 * read_lo, need_hi = std::min(2 * read_hi, S.h[s]);
            if (hi[s] > lo[s]) { lo[s] = std::min(lo[s], need_lo); hi[s] = std::max(hi[s], need_hi); }
            else { lo[s] = need_lo; hi[s] = need_hi; }
        }
        lo[s] &= ~1; // the tile grid starts on an even row (2x2 boxes)
        read_lo = std::max(lo[s] - 2, 0);
        read_hi = hi[s] > lo[s] ? std::min(hi[s] + 2, S.h[s]) : read_lo;
    }

    for (int s = 0; s < n_scales; s++) {
        if (hi[s] <= lo[s]) continue;
        LevelArgs A = {};
        A.w = S.w[s]; A.h = S.h[s];
        A.cover_lo = lo[s];
        A.rd_lo = std::max(lo[s] - 2, 0);
        A.rd_hi = std::min(hi[s] + 2, S.h[s]);
        A.y0 = S.y0[s]; A.y1 = S.y1[s];
        A.map = S.map[s];
        A.sum = S.d_sums + (size_t)s * kSlots;
        A.lut = S.d_lut;
        if (s + 1 < n_scales) {
            for (int i = 0; i < 2; i++)
                for (int c = 0; c < 3; c++) A.nxt[i][c] = S.lin[s + 1][i][c];
            A.nw = S.w[s + 1]; A.nh = S.h[s + 1];
            // rows of level s + 1 its own launch reads: its cover +- 2
            A.ny_lo = hi[s + 1] > lo[s + 1] ? std::max(lo[s + 1] - 2, 0) : 0;
            A.ny_hi = hi[s + 1] > lo[s + 1] ? std::min(hi[s + 1] + 2, S.h[s + 1]) : 0;
        }
        if (s == 0) {
            const int bpp = fr[0]->format == MVFX_FORMAT_RGBA ? 4 : 3;
            bool wide = bpp == 4;
            for (int i = 0; i < 2; i++) {
                A.bytes[i] = static_cast<const uint8_t *>(fr[i]->data);
                A.stride[i] = fr[i]->stride;
                wide = wide && ((reinterpret_cast<uintptr_t>(fr[i]->data) | fr[i]->stride) & 3) == 0;
            }
            bool vec8 = wide;
            for (int i = 0; i < 2; i++) vec8 = vec8 && ((reinterpret_cast<uintptr_t>(fr[i]->data) | fr[i]->stride) & 7) == 0;
            A.vec8 = vec8 ? 1 : 0;
            if (wide) launch_level<0, 4>(A, hi[s], st);
            else if (bpp == 4) launch_level<1, 4>(A, hi[s], st);
            else launch_level<1, 3>(A, hi[s], st);
        } else {
            for (int i = 0; i < 2; i++)
                for (int c = 0; c < 3; c++) A.lin[i][c] = S.lin[s][i][c];
            A.vec8 = (A.w & 1) == 0 ? 1 : 0; // hipMalloc'ed planes with an even pitch: (y w + x) even for even x
            launch_level<2, 4>(A, hi[s], st);
        }
    }
    S.scales = 0; // a pass 1 that fails below leaves nothing pending
    MVFX_HIP_TRY(hipGetLastError());
    double sums[kScales];
    if (int rc = read_slots(S.d_sums, sums, st); rc != MVFX_OK) return rc;
    S.scales = n_scales; // pass 2 of THIS pipeline is what mvfx_ssim_partial_deviation runs next on this thread
    for (int s = 0; s < kScales; s++) { // the device sums are those of the deficit 1 - ssim; the interface speaks of the map
        counts_out[s] = s < S.scales ? (double)S.w[s] * (double)std::max(S.y1[s] - S.y0[s], 0) : 0.0;
        sums_out[s] = s < S.scales ? counts_out[s] - sums[s] : 0.0;
    }
    *n_scales_out = (uint32_t)S.scales;
    return MVFX_OK;
}

bool pending() { return t_state.scales != 0; }
void abandon() { t_state.scales = 0; }

int partial_deviation(const double mean[5], double deviation_sums_out[5], hipStream_t st)
{
    State &S = t_state;
    DevArgs D = {};
    D.sum = S.d_sums + (size_t)kScales * kSlots;
    size_t most = 0;
    for (int s = 0; s < S.scales; s++) {
        D.map[s] = S.map[s];
        D.first[s] = (size_t)S.y0[s] * S.w[s];
        D.count[s] = S.y1[s] > S.y0[s] ? (size_t)S.w[s] * (size_t)(S.y1[s] - S.y0[s]) : 0;
        D.avg_deficit[s] = 1.0 - mean[s];
        most = std::max(most, D.count[s]);
    }
    if (most && S.scales) {
        const unsigned grid = (unsigned)std::min<size_t>((most + kThreads * 8 - 1) / (kThreads * 8), 2048);
        MVFX_LAUNCH(ssim32_dev_kernel, dim3(grid ? grid : 1, S.scales), dim3(kThreads), 0, st, D);
    }
    MVFX_HIP_TRY(hipGetLastError());
    double sums[kScales];
    if (int rc = read_slots(S.d_sums + (size_t)kScales * kSlots, sums, st); rc != MVFX_OK) return rc;
    for (int s = 0; s < kScales; s++)
        deviation_sums_out[s] = s < S.scales ? sums[s] : 0.0;
    S.scales = 0;
    return MVFX_OK;
}

} // namespace ssim32
} // namespace mvfx
