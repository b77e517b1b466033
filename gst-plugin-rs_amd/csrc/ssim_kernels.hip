// SSIM-family distance for videocompare's `hash-algo=dssim` on gfx950.
//
// The reference delegates to dssim-core 3.4.0 behind the NON-DEFAULT cargo feature `dssim`
// (video/videofx/Cargo.toml:39; call sites videocompare/hashed_image.rs:49-59,72-75); that crate
// is not under /root/reference.  This file implements the published structure of the algorithm as
// recorded in SURVEY.md Appendix A.3 (sRGB -> linear -> Lab-like planes, 5-level 2x box pyramid,
// binomial blur, per-scale SSIM map, mean adjusted by mean absolute deviation, fixed scale
// weights, 1/ssim - 1).  PARITY UNPINNED against the crate: the value is checked against this
// repository's own f64 restatement (oracle/ssim_oracle.c) and against the one property the
// reference's test pins (identical frames => 0, tests/videocompare.rs:141-182).
//
// Round 3: the DEFAULT path is the f32 pipeline of ssim32_kernels.hip (dssim-core is an f32 library; one fused kernel per
// pyramid level).  The kernels in THIS file are the f64 twin of round 2, selected per thread with MVFX_OPT_SSIM_F64:
// everything is f64 on the device (planes, window sums, reductions), within 1e-9 of the oracle -- the checker's twin.  The map of each
// scale is kept in device scratch between the two reduction passes (mean, then mean absolute
// deviation), so a multi-GPU caller can all-reduce the five partial sums in between
// (row bands with a 2-row halo per scale are read from the full frames resident on each GPU).
#include "mvfx_internal.h"
#include "ssim32.h"

#include <cmath>
#include <cstring>
#include <vector>

// The library is built with -ffp-contract=off for the bit-exact u8 paths; this file is a tolerance-checked
// f64 metric, so multiply-adds may fuse here.
#pragma clang fp contract(fast)

namespace mvfx {
namespace {

constexpr int kScales = 5;
const double kWeights[kScales] = {0.028, 0.197, 0.322, 0.298, 0.155};
constexpr double kC1 = 0.01 * 0.01, kC2 = 0.03 * 0.03;
constexpr int kBlock = 256;
constexpr int kSlots = 256; // accumulators per (pass, scale): one f64 atomic per workgroup, spread so they do not serialise

struct Planes {
    double *p[3];
    int w, h;
};

// All kernels work on the row range [y0, y0 + gridDim.y) of absolutely indexed full-size planes: a rank that
// owns a row band only fills the band plus the halo the coarser scales and the 5x5 window need.
// Full-resolution linear RGB is never stored: scale 0 goes straight from the bytes to the Lab planes, scale 1
// averages the four source pixels itself, and every coarser level writes its linear planes (for the next
// level) and its Lab planes (for the map) in one pass -- 1.8 GB instead of 4.3 GB of f64 traffic per 8K image.
// cbrt(t) for t in (216/24389, ~1]: t^(-1/3) from the f32 log2 / exp2 units (relative error < 1e-6), two division-free
// Newton steps r <- r (4 - t r^3) / 3 in f64 (1e-6 -> 2e-12 -> < 1e-20), cbrt = t r^2: ~15 instructions against the ~70 of the
// library routine, within 2 ulp of it (the conversion kernels were VALU bound: 243 instructions per pixel, 75 % VALU busy,
// three cube roots and five divisions by constants per pixel; round 2).
__device__ __forceinline__ double cbrt_unit(double t)
{
    double r = (double)__builtin_amdgcn_exp2f(-0.33333334f * __builtin_amdgcn_logf((float)t));
    const double t3 = t * (1.0 / 3.0);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double r3 = r * r * r;
        r = r * (4.0 / 3.0 - t3 * r3);
    }
    return t * (r * r);
}

__device__ __forceinline__ double lab_f(double t)
{
    const double eps = 216.0 / 24389.0, kappa = 24389.0 / 27.0;
    return t > eps ? cbrt_unit(t) : (kappa * t + 16.0) * (1.0 / 116.0);
}

__device__ __forceinline__ void store_lab(const Planes &lab, size_t i, double r, double g, double b)
{
    // the oracle divides by 0.9505, 1.089, 100 and 220; multiplying by the reciprocals differs by <= 1 ulp per operation
    const double X = (0.4124 * r + 0.3576 * g + 0.1805 * b) * (1.0 / 0.9505);
    const double Y = 0.2126 * r + 0.7152 * g + 0.0722 * b;
    const double Z = (0.0193 * r + 0.1192 * g + 0.9505 * b) * (1.0 / 1.089);
    const double fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    lab.p[0][i] = (116.0 * fy - 16.0) * (1.0 / 100.0);
    lab.p[1][i] = (86.2 + 500.0 * (fx - fy)) * (1.0 / 220.0);
    lab.p[2][i] = (107.9 + 200.0 * (fy - fz)) * (1.0 / 220.0);
}

// linear RGB of one source pixel, alpha premultiplied.  `lut` (sRGB byte -> linear f64) is the workgroup's LDS copy; WIDE:
// 4-byte pixels in 4-byte aligned rows are fetched as one dword.  (Round 2: with four byte loads and three table loads from
// global memory per pixel the two conversion kernels were bound by the vector L1's tag rate -- 1.4e8 accesses per 8K launch,
// 0.55 M per CU at ~1 per clock = 0.23 of the 0.28 ms.)
template <int BPP, bool WIDE>
__device__ __forceinline__ void linear_px(const uint8_t *p, const double *lut, double &r, double &g, double &b)
{
    uint32_t c0, c1, c2, c3 = 255;
    if (WIDE) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
        c0 = v & 0xffu; c1 = (v >> 8) & 0xffu; c2 = (v >> 16) & 0xffu; c3 = v >> 24;
    } else {
        c0 = p[0]; c1 = p[1]; c2 = p[2];
        if (BPP == 4) c3 = p[3];
    }
    const double a = BPP == 4 ? c3 * (1.0 / 255.0) : 1.0; // 255 * (1/255) == 1 exactly
    r = lut[c0] * a; g = lut[c1] * a; b = lut[c2] * a;
}

// scale 0: bytes -> Lab
template <int BPP, bool WIDE>
__global__ __launch_bounds__(kBlock) void ssim_lab0_kernel(const uint8_t *frame, int y0, uint64_t stride, const double *lut, Planes lab)
{
    __shared__ double s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * kBlock + threadIdx.x, y = y0 + blockIdx.y;
    if (x >= lab.w) return;
    double r, g, b;
    linear_px<BPP, WIDE>(frame + (uint64_t)y * stride + (uint64_t)x * BPP, s_lut, r, g, b);
    store_lab(lab, (size_t)y * lab.w + x, r, g, b);
}

// scale 1: 2x2 box of the linearised source pixels -> linear planes + Lab planes of the half-size image
template <int BPP, bool WIDE>
__global__ __launch_bounds__(kBlock) void ssim_down1_kernel(const uint8_t *frame, int y0, uint64_t stride, const double *lut, Planes lin,
                                                            Planes lab)
{
    __shared__ double s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * kBlock + threadIdx.x, y = y0 + blockIdx.y;
    if (x >= lin.w) return;
    const uint8_t *r0 = frame + (uint64_t)(2 * y) * stride + (uint64_t)(2 * x) * BPP, *r1 = r0 + stride;
    double v[4][3];
    linear_px<BPP, WIDE>(r0, s_lut, v[0][0], v[0][1], v[0][2]);
    linear_px<BPP, WIDE>(r0 + BPP, s_lut, v[1][0], v[1][1], v[1][2]);
    linear_px<BPP, WIDE>(r1, s_lut, v[2][0], v[2][1], v[2][2]);
    linear_px<BPP, WIDE>(r1 + BPP, s_lut, v[3][0], v[3][1], v[3][2]);
    double o[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
        o[c] = (v[0][c] + v[1][c] + v[2][c] + v[3][c]) * 0.25;
    const size_t i = (size_t)y * lin.w + x;
    lin.p[0][i] = o[0]; lin.p[1][i] = o[1]; lin.p[2][i] = o[2];
    store_lab(lab, i, o[0], o[1], o[2]);
}

// scale >= 2: 2x2 box of the previous level's linear planes -> this level's linear + Lab planes
__global__ __launch_bounds__(kBlock) void ssim_downlab_kernel(Planes in, Planes lin, Planes lab, int y0)
{
    const int x = blockIdx.x * kBlock + threadIdx.x, y = y0 + blockIdx.y;
    if (x >= lin.w) return;
    double o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double *r0 = in.p[c] + (size_t)(2 * y) * in.w + 2 * x, *r1 = r0 + in.w;
        o[c] = (r0[0] + r0[1] + r1[0] + r1[1]) * 0.25;
    }
    const size_t i = (size_t)y * lin.w + x;
    lin.p[0][i] = o[0]; lin.p[1][i] = o[1]; lin.p[2][i] = o[2];
    store_lab(lab, i, o[0], o[1], o[2]);
}

__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double part[kBlock / 64];
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kBlock / 64; i++) t += part[i];
    return t; // valid in thread 0
}

// SSIM map of rows [y0,y1) of one scale + its sum.  The 5x5 binomial window is separable: a workgroup owns a
// kTileW x kSeg tile.  Per channel: the raw values of both images for the tile + 2-pixel halo go into LDS ONCE (R);
// pass A forms the five products per pixel and their horizontal blur for two neighbouring columns per job (six raw values of
// each image = three 16-byte LDS reads) into H; pass B runs the vertical blur down each column from H and the SSIM term, while
// the next channel's raw values are already on their way from memory (registers -> R after the pass).
// History: 25-tap window per pixel from global memory 1.7 ms for the 8K scale-0 map; separable with pass A reading its ten
// taps per entry from global memory 0.95 ms -- that version made 5e8 vector-L1 accesses per 8K pair (2 M per CU at ~1 per
// clock = 0.8 of its 1.28 ms, VALU 27 % busy; profiles/r2/ssim_counters_before.txt): the L1 tag rate was the bound.
template <int kSeg, int kTileW>
__global__ __launch_bounds__(kBlock) void ssim_map_kernel(Planes a, Planes b, int y0, int y1, double *map, double *sum)
{
    constexpr int kRowsPerLane = kSeg * kTileW / kBlock;
    static_assert(kRowsPerLane * kBlock == kSeg * kTileW && kBlock % kTileW == 0 && kTileW % 2 == 0, "tile shape");
    constexpr int kRows = kSeg + 4, kRawCols = kTileW + 4, kRawStride = kTileW + 6; // stride: even (16-byte pairs), not a multiple of 32
    constexpr int kRawN = kRows * kRawCols;
    constexpr int kRawPerLane = (2 * kRawN + kBlock - 1) / kBlock;
    constexpr int kPairs = kRows * (kTileW / 2);
    __shared__ __attribute__((aligned(16))) double R[2][kRows][kRawStride];
    __shared__ __attribute__((aligned(16))) double H[5][kRows][kTileW];
    const int w = a.w, h = a.h;
    const int tx0 = blockIdx.x * kTileW, ty0 = y0 + blockIdx.y * kSeg;
    const double B0 = 1.0 / 16, B1 = 4.0 / 16, B2 = 6.0 / 16;
    const int col = threadIdx.x % kTileW, rg = threadIdx.x / kTileW; // pass B: column, group of rows
    double acc[kRowsPerLane] = {};

    // raw job j -> (image, row, column) of the haloed tile and its clamped source index (the same for every channel)
    int raw_lds[kRawPerLane];
    size_t raw_src[kRawPerLane];
#pragma unroll
    for (int k = 0; k < kRawPerLane; k++) {
        const int j = min((int)threadIdx.x + k * kBlock, 2 * kRawN - 1);
        const int img = j / kRawN, rem = j % kRawN, row = rem / kRawCols, rc = rem % kRawCols;
        const int yy = min(max(ty0 - 2 + row, 0), h - 1), xx = min(max(tx0 - 2 + rc, 0), w - 1);
        raw_lds[k] = (img * kRows + row) * kRawStride + rc;
        raw_src[k] = (size_t)yy * w + xx;
    }
    double nxt[kRawPerLane];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < kRawPerLane; k++) {
            const bool second = (int)threadIdx.x + k * kBlock >= kRawN;
            nxt[k] = (second ? b.p[c] : a.p[c])[raw_src[k]];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < kRawPerLane; k++)
            if ((int)threadIdx.x + k * kBlock < 2 * kRawN) (&R[0][0][0])[raw_lds[k]] = nxt[k];
    };
    fetch(0);
    stage();
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        for (int p = threadIdx.x; p < kPairs; p += kBlock) {
            const int row = p / (kTileW / 2), cc = (p % (kTileW / 2)) * 2;
            double v1[6], v2[6];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double2 t1 = *reinterpret_cast<const double2 *>(&R[0][row][cc + 2 * k]);
                const double2 t2 = *reinterpret_cast<const double2 *>(&R[1][row][cc + 2 * k]);
                v1[2 * k] = t1.x; v1[2 * k + 1] = t1.y;
                v2[2 * k] = t2.x; v2[2 * k + 1] = t2.y;
            }
            double q[5][6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                q[0][k] = v1[k]; q[1][k] = v2[k];
                q[2][k] = v1[k] * v1[k]; q[3][k] = v2[k] * v2[k]; q[4][k] = v1[k] * v2[k];
            }
#pragma unroll
            for (int m = 0; m < 5; m++) {
                double2 o;
                o.x = B0 * (q[m][0] + q[m][4]) + B1 * (q[m][1] + q[m][3]) + B2 * q[m][2];
                o.y = B0 * (q[m][1] + q[m][5]) + B1 * (q[m][2] + q[m][4]) + B2 * q[m][3];
                *reinterpret_cast<double2 *>(&H[m][row][cc]) = o;
            }
        }
        __syncthreads();
        if (c < 2) fetch(c + 1); // R is free from here on; the loads fly during pass B
#pragma unroll
        for (int r = 0; r < kRowsPerLane; r++) {
            const int tr = rg * kRowsPerLane + r; // output row of the tile; window rows tr .. tr + 4 of H
            double m[5];
#pragma unroll
            for (int q = 0; q < 5; q++)
                m[q] = B0 * (H[q][tr][col] + H[q][tr + 4][col]) + B1 * (H[q][tr + 1][col] + H[q][tr + 3][col]) + B2 * H[q][tr + 2][col];
            const double m1 = m[0], m2 = m[1];
            const double s11 = m[2] - m1 * m1, s22 = m[3] - m2 * m2, s12 = m[4] - m1 * m2;
            acc[r] += ((2.0 * m1 * m2 + kC1) * (2.0 * s12 + kC2)) / ((m1 * m1 + m2 * m2 + kC1) * (s11 + s22 + kC2));
        }
        if (c < 2) stage();
        __syncthreads();
    }
    double total = 0.0;
    const int x = tx0 + col;
#pragma unroll
    for (int r = 0; r < kRowsPerLane; r++) {
        const int y = ty0 + rg * kRowsPerLane + r;
        if (x < w && y < y1) {
            const double val = acc[r] / 3.0;
            map[(size_t)y * w + x] = val;
            total += val;
        }
    }
    const double t = block_sum(total);
    if (threadIdx.x == 0) atomicAdd(sum + ((blockIdx.x + blockIdx.y * gridDim.x) % kSlots), t);
}

__global__ __launch_bounds__(kBlock) void ssim_dev_kernel(const double *map, int w, int y0, int y1, double avg, double *sum)
{
    const int x = blockIdx.x * kBlock + threadIdx.x, y = y0 + blockIdx.y;
    double val = 0.0;
    if (x < w && y < y1)
        val = fabs(map[(size_t)y * w + x] - avg);
    const double t = block_sum(val);
    if (threadIdx.x == 0) atomicAdd(sum + ((blockIdx.x + blockIdx.y * gridDim.x) % kSlots), t);
}

// Per-thread scratch, kept across calls while the frame size stays the same (a videocompare pad pair per
// buffer): f64 planes of both images (linear RGB, ping-pong for the pyramid, Lab) and the five maps.
struct SsimState {
    std::vector<void *> allocations;
    int w0 = 0, h0 = 0, device = -1;
    Planes half[2], quarter[2], lab[2]; // linear RGB of the odd / even coarser scales (ping-pong), Lab of the current scale
    double *map[kScales] = {};
    int w[kScales] = {}, h[kScales] = {}, y0[kScales] = {}, y1[kScales] = {};
    int scales = 0;        // of the last mvfx_ssim_partial_sums on this thread (0: none pending)
    double *d_sums = nullptr; // [2 passes][kScales][kSlots] doubles
    double *d_lut = nullptr;
    void release()
    {
        for (void *p : allocations) (void)hipFree(p);
        allocations.clear();
        scales = 0;
        w0 = h0 = 0;
        d_sums = nullptr;
        d_lut = nullptr;
    }
    ~SsimState() { release(); }
};
thread_local SsimState t_ssim;

int dalloc(SsimState &st, size_t bytes, void **out)
{
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "ssim: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    st.allocations.push_back(*out);
    return MVFX_OK;
}

int alloc_planes(SsimState &st, int w, int h, Planes *pl)
{
    pl->w = w; pl->h = h;
    for (int c = 0; c < 3; c++)
        if (int rc = dalloc(st, sizeof(double) * (size_t)w * h, reinterpret_cast<void **>(&pl->p[c])); rc != MVFX_OK) return rc;
    return MVFX_OK;
}

// (Re)allocates the scratch of this thread for w0 x h0 frames on the current device.
int ensure_scratch(SsimState &S, int w0, int h0, hipStream_t st)
{
    int dev = 0;
    MVFX_HIP_TRY(hipGetDevice(&dev));
    if (S.w0 == w0 && S.h0 == h0 && S.device == dev && !S.allocations.empty())
        return MVFX_OK;
    S.release();
    S.device = dev;
    if (int rc = dalloc(S, sizeof(double) * 2 * kScales * kSlots, reinterpret_cast<void **>(&S.d_sums)); rc != MVFX_OK) return rc;
    if (int rc = dalloc(S, sizeof(double) * 256, reinterpret_cast<void **>(&S.d_lut)); rc != MVFX_OK) return rc;
    double lut[256];
    for (int i = 0; i < 256; i++) {
        const double x = i / 255.0;
        lut[i] = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
    }
    MVFX_HIP_TRY(hipMemcpyAsync(S.d_lut, lut, sizeof(lut), hipMemcpyHostToDevice, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st)); // `lut` is a stack buffer
    for (int i = 0; i < 2; i++) {
        if (int rc = alloc_planes(S, w0, h0, &S.lab[i]); rc != MVFX_OK) return rc;
        if (int rc = alloc_planes(S, w0 / 2, h0 / 2, &S.half[i]); rc != MVFX_OK) return rc;
        if (int rc = alloc_planes(S, std::max(w0 / 4, 1), std::max(h0 / 4, 1), &S.quarter[i]); rc != MVFX_OK) return rc;
    }
    int w = w0, h = h0;
    for (int s = 0; s < kScales; s++) {
        if (s > 0) {
            if (w / 2 < 8 || h / 2 < 8) break;
            w /= 2; h /= 2;
        }
        if (int rc = dalloc(S, sizeof(double) * (size_t)w * h, reinterpret_cast<void **>(&S.map[s])); rc != MVFX_OK) return rc;
    }
    S.w0 = w0; S.h0 = h0;
    return MVFX_OK;
}

template <int kSeg, int kTileW>
void launch_map_t(int w, int y0, int y1, hipStream_t st, const Planes &a, const Planes &b, double *map, double *sum)
{
    MVFX_LAUNCH((ssim_map_kernel<kSeg, kTileW>), dim3((w + kTileW - 1) / kTileW, (y1 - y0 + kSeg - 1) / kSeg), dim3(kBlock), 0, st,
                       a, b, y0, y1, map, sum);
}

// Tile shape, 8K pair end to end: 64x16 2.24 ms, 32x16 2.16 ms (38 KB of LDS: four workgroups per CU), 32x32 2.24 ms, 64x32 2.49 ms.
void launch_map(int w, int y0, int y1, hipStream_t st, const Planes &a, const Planes &b, double *map, double *sum)
{
    launch_map_t<16, 32>(w, y0, y1, st, a, b, map, sum);
}

dim3 grid2d(int w, int rows) { return dim3((w + kBlock - 1) / kBlock, rows > 0 ? rows : 1, 1); }

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_ssim_partial_sums(const mvfx_frame *reference_frame, const mvfx_frame *other_frame, uint32_t row_begin,
                           uint32_t row_end, double sums_out[5], double counts_out[5], uint32_t *n_scales_out,
                           mvfx_stream stream)
{
    if (!reference_frame || !other_frame || !sums_out || !counts_out || !n_scales_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: NULL argument");
    const mvfx_frame *fr[2] = {reference_frame, other_frame};
    for (const mvfx_frame *f : fr) {
        if (f->format != MVFX_FORMAT_RGB && f->format != MVFX_FORMAT_RGBA)
            return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "videocompare: format %d is not RGB / RGBA (videocompare/imp.rs:160-162)", f->format);
        if (int rc = check_packed_frame(f, "videocompare"); rc != MVFX_OK) return rc;
    }
    if (reference_frame->width != other_frame->width || reference_frame->height != other_frame->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "Video streams do not have the same sizes (videocompare/imp.rs:337-346)");
    const int w0 = (int)reference_frame->width, h0 = (int)reference_frame->height;
    if (w0 < 8 || h0 < 8)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: frames smaller than 8x8 are not supported");
    if (row_end > (uint32_t)h0) row_end = (uint32_t)h0;
    if (row_begin > row_end) row_begin = row_end;
    if ((row_begin % 16) != 0 || (row_end != (uint32_t)h0 && (row_end % 16) != 0))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: band boundaries must be multiples of 16 rows (5 pyramid levels)");
    if (int rc = require_device(); rc != MVFX_OK) return rc;

    hipStream_t st = as_stream(stream);
    // whichever pipeline runs THIS pass 1 is the one mvfx_ssim_partial_deviation continues: a pass 1 of the other pipeline that never
    // got its pass 2 (a failed read, a caller that gave the pair up) must not be picked up later with stale maps
    if ((thread_options() & MVFX_OPT_SSIM_F64) == 0) { // default: the f32 pipeline (what dssim-core computes in); f64 planes on request
        t_ssim.scales = 0;
        return ssim32::partial_sums(fr, row_begin, row_end, sums_out, counts_out, n_scales_out, st);
    }
    ssim32::abandon();
    SsimState &S = t_ssim;
    S.scales = 0;
    if (int rc = ensure_scratch(S, w0, h0, st); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemsetAsync(S.d_sums, 0, sizeof(double) * 2 * kScales * kSlots, st));

    // geometry of every scale, the band's rows [y0,y1) at that scale, and the rows [a,b) of the planes that must
    // exist there: the band +-2 (5x5 window) and twice the range the next coarser scale needs
    int ws[kScales], hs[kScales], a[kScales], b[kScales], n_scales = 0;
    for (int s = 0, w = w0, h = h0; s < kScales; s++) {
        if (s > 0) {
            if (w / 2 < 8 || h / 2 < 8) break;
            w /= 2; h /= 2;
        }
        ws[s] = w; hs[s] = h;
        S.w[s] = w; S.h[s] = h;
        S.y0[s] = std::min((int)(row_begin >> s), h);
        S.y1[s] = row_end == (uint32_t)h0 ? h : std::min((int)(row_end >> s), h);
        n_scales = s + 1;
    }
    for (int s = n_scales - 1; s >= 0; s--) {
        a[s] = std::max(S.y0[s] - 2, 0);
        b[s] = std::min(S.y1[s] + 2, hs[s]);
        if (S.y1[s] <= S.y0[s]) { a[s] = 0; b[s] = 0; }
        if (s + 1 < n_scales && b[s + 1] > a[s + 1]) {
            const int lo = 2 * a[s + 1], hi = std::min(2 * b[s + 1], hs[s]);
            if (b[s] > a[s]) { a[s] = std::min(a[s], lo); b[s] = std::max(b[s], hi); }
            else { a[s] = lo; b[s] = hi; }
        }
    }

    for (int s = 0; s < n_scales; s++) {
        const int w = ws[s], h = hs[s];
        Planes lab[2];
        if (b[s] > a[s])
            for (int i = 0; i < 2; i++) {
                const int bpp = fr[i]->format == MVFX_FORMAT_RGBA ? 4 : 3;
                const uint8_t *src = static_cast<const uint8_t *>(fr[i]->data);
                lab[i] = S.lab[i];
                lab[i].w = w; lab[i].h = h;
                Planes out = (s & 1) ? S.half[i] : S.quarter[i]; // scale 1, 3 -> half-size buffer; 2, 4 -> quarter-size buffer
                out.w = w; out.h = h;
                const dim3 grid = grid2d(w, b[s] - a[s]);
                const uint64_t stride = fr[i]->stride;
                const bool wide = bpp == 4 && ((reinterpret_cast<uintptr_t>(src) | stride) & 3) == 0;
                if (s == 0) {
                    if (wide) MVFX_LAUNCH((ssim_lab0_kernel<4, true>), grid, dim3(kBlock), 0, st, src, a[0], stride, S.d_lut, lab[i]);
                    else if (bpp == 4) MVFX_LAUNCH((ssim_lab0_kernel<4, false>), grid, dim3(kBlock), 0, st, src, a[0], stride, S.d_lut, lab[i]);
                    else MVFX_LAUNCH((ssim_lab0_kernel<3, false>), grid, dim3(kBlock), 0, st, src, a[0], stride, S.d_lut, lab[i]);
                } else if (s == 1) {
                    if (wide) MVFX_LAUNCH((ssim_down1_kernel<4, true>), grid, dim3(kBlock), 0, st, src, a[1], stride, S.d_lut, out, lab[i]);
                    else if (bpp == 4) MVFX_LAUNCH((ssim_down1_kernel<4, false>), grid, dim3(kBlock), 0, st, src, a[1], stride, S.d_lut, out, lab[i]);
                    else MVFX_LAUNCH((ssim_down1_kernel<3, false>), grid, dim3(kBlock), 0, st, src, a[1], stride, S.d_lut, out, lab[i]);
                } else {
                    Planes in = (s & 1) ? S.quarter[i] : S.half[i];
                    in.w = ws[s - 1]; in.h = hs[s - 1];
                    MVFX_LAUNCH(ssim_downlab_kernel, grid, dim3(kBlock), 0, st, in, out, lab[i], a[s]);
                }
            }
        if (S.y1[s] > S.y0[s])
            launch_map(w, S.y0[s], S.y1[s], st, lab[0], lab[1], S.map[s], S.d_sums + (size_t)s * kSlots);
    }
    S.scales = n_scales;
    MVFX_HIP_TRY(hipGetLastError());
    std::vector<double> slots((size_t)kScales * kSlots);
    MVFX_HIP_TRY(hipMemcpyAsync(slots.data(), S.d_sums, slots.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    double sums[kScales];
    for (int s = 0; s < kScales; s++) {
        sums[s] = 0.0;
        for (int k = 0; k < kSlots; k++) sums[s] += slots[(size_t)s * kSlots + k];
    }
    for (int s = 0; s < kScales; s++) {
        sums_out[s] = s < S.scales ? sums[s] : 0.0;
        counts_out[s] = s < S.scales ? (double)S.w[s] * (double)std::max(S.y1[s] - S.y0[s], 0) : 0.0;
    }
    *n_scales_out = (uint32_t)S.scales;
    return MVFX_OK;
}

int mvfx_ssim_partial_deviation(const double mean[5], double deviation_sums_out[5], mvfx_stream stream)
{
    if (!mean || !deviation_sums_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: NULL argument");
    if (ssim32::pending())
        return ssim32::partial_deviation(mean, deviation_sums_out, as_stream(stream));
    SsimState &S = t_ssim;
    if (S.scales == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: mvfx_ssim_partial_sums must be called first on this thread");
    hipStream_t st = as_stream(stream);
    for (int s = 0; s < S.scales; s++)
        if (S.y1[s] > S.y0[s])
            MVFX_LAUNCH(ssim_dev_kernel, grid2d(S.w[s], S.y1[s] - S.y0[s]), dim3(kBlock), 0, st, S.map[s], S.w[s], S.y0[s],
                               S.y1[s], mean[s], S.d_sums + (kScales + s) * kSlots);
    MVFX_HIP_TRY(hipGetLastError());
    std::vector<double> slots((size_t)kScales * kSlots);
    MVFX_HIP_TRY(hipMemcpyAsync(slots.data(), S.d_sums + (size_t)kScales * kSlots, slots.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    double sums[kScales];
    for (int s = 0; s < kScales; s++) {
        sums[s] = 0.0;
        for (int k = 0; k < kSlots; k++) sums[s] += slots[(size_t)s * kSlots + k];
    }
    for (int s = 0; s < kScales; s++)
        deviation_sums_out[s] = s < S.scales ? sums[s] : 0.0;
    S.scales = 0; // the maps are consumed; the scratch stays for the next pair of the same size
    return MVFX_OK;
}

double mvfx_ssim_combine(const double mean[5], const double mean_abs_deviation[5], uint32_t n_scales)
{
    double num = 0.0, den = 0.0;
    for (uint32_t s = 0; s < n_scales && s < (uint32_t)kScales; s++) {
        num += kWeights[s] * (mean[s] - mean_abs_deviation[s]);
        den += kWeights[s];
    }
    const double ssim = den > 0 ? num / den : 1.0;
    return 1.0 / (ssim > 1e-12 ? ssim : 1e-12) - 1.0;
}

int mvfx_ssim_distance(const mvfx_frame *reference_frame, const mvfx_frame *other_frame, double *distance_out, mvfx_stream stream)
{
    if (!distance_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: NULL output");
    double sums[5], counts[5], mean[5], dev[5];
    uint32_t n = 0;
    if (int rc = mvfx_ssim_partial_sums(reference_frame, other_frame, 0, reference_frame ? reference_frame->height : 0, sums,
                                        counts, &n, stream); rc != MVFX_OK) return rc;
    for (uint32_t s = 0; s < n; s++) mean[s] = sums[s] / counts[s];
    if (int rc = mvfx_ssim_partial_deviation(mean, dev, stream); rc != MVFX_OK) return rc;
    for (uint32_t s = 0; s < n; s++) dev[s] /= counts[s];
    *distance_out = mvfx_ssim_combine(mean, dev, n);
    return MVFX_OK;
}

int mvfx_ssim_distance_host(const mvfx_frame *reference_frame, const mvfx_frame *other_frame, double *distance_out)
{
    if (!reference_frame || !other_frame)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "ssim: NULL frame");
    if (int rc = check_packed_frame(reference_frame, "videocompare"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(other_frame, "videocompare"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t ba = (size_t)reference_frame->stride * reference_frame->height, bb = (size_t)other_frame->stride * other_frame->height;
    void *da = nullptr, *db = nullptr;
    if (int rc = host_scratch(ba ? ba : 16, 0, &da); rc != MVFX_OK) return rc;
    if (int rc = host_scratch(bb ? bb : 16, 1, &db); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    if (ba) MVFX_HIP_TRY(hipMemcpyAsync(da, reference_frame->data, ba, hipMemcpyHostToDevice, st));
    if (bb) MVFX_HIP_TRY(hipMemcpyAsync(db, other_frame->data, bb, hipMemcpyHostToDevice, st));
    mvfx_frame fa = *reference_frame, fb = *other_frame;
    fa.data = da;
    fb.data = db;
    return mvfx_ssim_distance(&fa, &fb, distance_out, st);
}

} // extern "C"
