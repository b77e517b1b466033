// videofx kernels for gfx950: colordetect histogram, videocompare block sums, roundedcorners
// alpha mask + A420 compose; and their C ABI.
//
//  colordetect  (video/videofx/src/colordetect/imp.rs:57-86 -> color_thief::get_palette)
//     The O(pixels) part of MMCQ is the 5-5-5 histogram over every `quality`-th pixel of the
//     flat plane (padding included, colordetect/imp.rs:69).  Kernel: 1024-thread workgroups,
//     LDS-privatised 32768-bin histogram packed as 16-bit pairs (64 KiB; each workgroup is given
//     < 65536 samples so a 16-bit bin cannot overflow), per-wave min/max folded through LDS, the
//     packed histogram of every workgroup written out densely (no global atomics: content
//     independent timing) and a second launch that adds the partials.  The serial median cut runs
//     on the host (host/mmcq.cpp), as it does in the reference.
//  videocompare (video/videofx/src/videocompare/hashed_image.rs:24-79 -> image_hasher Blockhash)
//     64 block sums (u32) of r+g+b (765 when alpha==0) over an 8x8 grid of W/8 x H/8 blocks:
//     16-byte non-temporal reads (4 rows in flight per lane), wave shuffle reduction, one plain store per
//     workgroup, then a 64-wave launch that adds the partials (no same-address atomics, no memset);
//     every pad of an aggregate in one launch (blockIdx.z).
//     A row range can be given so that 8 ranks each reduce one block-row and all-reduce 64 u32.
//  roundedcorners (video/videofx/src/border/imp.rs:57-180, 482-559)
//     The A8 mask is rendered ONCE per caps / radius change by libcairo, exactly as the reference
//     does (host/cairo_mask.cpp replays border/imp.rs:64-103 through the system library), and
//     uploaded; the per-frame device work is the I420 -> A420 compose (plane copies + mask) for
//     device-resident pipelines.  radius 0 => 0xFF everywhere (border/imp.rs:123-128).
#include "mvfx_internal.h"

#include "cairo_mask.h"
#include "mmcq.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mvfx {
namespace {

// ------------------------------------------------------------------ colordetect

typedef uint32_t sum_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHistBlock = 1024;
constexpr uint32_t kHistWords = kHistBins / 2;         // packed u16 pairs
constexpr uint32_t kMaxSamplesPerGroup = 65535;        // 16-bit bins cannot overflow
constexpr uint32_t kMaxGroupsPerLaunch = 1024;          // 64 MiB of partials
constexpr int kHistInFlight = 8;                       // sample loads a lane issues before binning
constexpr uint32_t kPartialWords = kHistWords + 8;     // scratch per group: 16384 packed pairs + 6 bounds (+2 pad)

struct HistLayout {
    int bpp, ir, ig, ib, ia; // ia < 0: no alpha (255)
};

__global__ __launch_bounds__(kHistBlock) void colordetect_hist_kernel(
    const uint8_t *plane, uint64_t first_sample, uint64_t n_samples, uint32_t samples_per_group,
    uint32_t quality, HistLayout lay, uint32_t *partials)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist_lds[]; // kHistWords + 8 words (> 64 KiB: dynamic)
    uint32_t *bins = hist_lds;
    uint32_t *s_min = hist_lds + kHistWords, *s_max = hist_lds + kHistWords + 4;
    for (uint32_t i = threadIdx.x; i < kHistWords; i += kHistBlock)
        bins[i] = 0;
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 255; s_max[threadIdx.x] = 0; }
    __syncthreads();

    const uint64_t g_begin = (uint64_t)blockIdx.x * samples_per_group;
    const uint64_t g_end = min(g_begin + samples_per_group, n_samples);
    uint32_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    // kHistInFlight sample loads per lane are issued before the first one is binned.  4K, 1 / 2 / 4 / 8 in flight:
    // q=1 (64 samples per lane) 55.9 / 46.1 / 45.2 / 41.2 us; q=10 (6 samples per lane) 14.3 / 14.2 / 15.0 / 14.5 us on smooth
    // content, 22.9 / 22.0 / 22.6 / 22.3 us on uniform-random colours.  (A first attempt with the loads inside
    // `if (k < g_end)` was slower with every batch size: see the comment below.)
    const bool dword = lay.bpp == 4 && ((reinterpret_cast<uintptr_t>(plane) | (uintptr_t)(quality * 4u)) & 3) == 0;
    for (uint64_t k0 = g_begin + threadIdx.x; k0 < g_end; k0 += (uint64_t)kHistBlock * kHistInFlight) {
        uint32_t px[kHistInFlight];
        // unconditional loads from clamped sample indices, masked afterwards (a load inside a divergent branch gets its
        // own s_waitcnt and the loads of a batch would go one memory round trip after the other)
        if (dword) {
            uint32_t v[kHistInFlight];
#pragma unroll
            for (int j = 0; j < kHistInFlight; j++) {
                const uint64_t k = min(k0 + (uint64_t)j * kHistBlock, g_end - 1);
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(plane + (first_sample + k) * quality * 4ull));
            }
#pragma unroll
            for (int j = 0; j < kHistInFlight; j++) {
                const uint32_t q = ((v[j] >> (8 * lay.ir)) & 0xff) | (((v[j] >> (8 * lay.ig)) & 0xff) << 8) |
                                   (((v[j] >> (8 * lay.ib)) & 0xff) << 16) | (((v[j] >> (8 * lay.ia)) & 0xff) << 24);
                px[j] = k0 + (uint64_t)j * kHistBlock < g_end ? q : 0u; // alpha 0: skipped below
            }
        } else {
#pragma unroll
            for (int j = 0; j < kHistInFlight; j++) {
                const uint64_t k = min(k0 + (uint64_t)j * kHistBlock, g_end - 1);
                const uint8_t *p = plane + (first_sample + k) * quality * (uint64_t)lay.bpp;
                const uint32_t q = (uint32_t)p[lay.ir] | ((uint32_t)p[lay.ig] << 8) | ((uint32_t)p[lay.ib] << 16) |
                                   ((lay.ia >= 0 ? (uint32_t)p[lay.ia] : 255u) << 24);
                px[j] = k0 + (uint64_t)j * kHistBlock < g_end ? q : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < kHistInFlight; j++) {
            uint32_t r = px[j] & 0xff, g = (px[j] >> 8) & 0xff, b = (px[j] >> 16) & 0xff;
            const uint32_t a = px[j] >> 24;
            if (a < 125 || (r > 250 && g > 250 && b > 250)) // mostly transparent or white: skipped
                continue;
            r >>= 3; g >>= 3; b >>= 3;
            mn[0] = min(mn[0], r); mx[0] = max(mx[0], r);
            mn[1] = min(mn[1], g); mx[1] = max(mx[1], g);
            mn[2] = min(mn[2], b); mx[2] = max(mx[2], b);
            const uint32_t bin = (r << 10) | (g << 5) | b;
            atomicAdd(&bins[bin >> 1], 1u << ((bin & 1) * 16));
        }
    }
    // wave-level min/max, then one LDS atomic per wave
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[c] = min(mn[c], (uint32_t)__shfl_down((int)mn[c], off));
            mx[c] = max(mx[c], (uint32_t)__shfl_down((int)mx[c], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_min[c], mn[c]);
            atomicMax(&s_max[c], mx[c]);
        }
    }
    __syncthreads();
    // flush: the group's packed histogram goes out DENSE (64 KiB of coalesced 16-byte stores, no atomics) and
    // colordetect_reduce_kernel adds the groups' partials.  History: round 1 flushed the non-empty words with device-scope
    // 64-bit atomics into one table (~55 G atomics/s: 15 us for the 0.8 M non-empty words of a uniform-random 4K frame at
    // quality 10); round 2 first moved them into per-XCD tables with workgroup-scope atomics (they execute in the XCD's own
    // L2; q=1 56 -> 43.6 us, q=10 random unchanged at 24 us: ~85 G atomics/s is still ~10 us for 0.8 M of them), then
    // dropped the atomics: the partials are G x 64 KiB = 12 MiB at 4K, written (non-temporal) and read once.  4K quality 10,
    // both launches: 26.5 us (uniform-random colours) / 13.9 us (smooth content) with atomics -> 15.8 us whatever the
    // content; quality 1: 48.5 -> 27.3 us.  (An earlier dense attempt -- row-major partials, a 32-workgroup summing launch --
    // took 37 us: the layout below and a full-width second launch are what make it pay.)
    // Phase experiment (4K, quality 10, 10.6 us for this launch): without the partial store 9.8 us, without the LDS atomics 10.4,
    // without the sample loads 9.7, without all three 5.4, additionally without the LDS clear 4.7 us -- the phases hide behind
    // each other and half of the launch is the fixed cost of starting 192 x 1024 lanes with 64 KiB of LDS each; the second launch
    // costs 4.0 us with nothing to add up.
    // layout: [block of 16 uint4 columns][group][16 uint4]: the reduce kernel's workgroup (one column block) reads the
    // groups' 256-byte pieces as ONE contiguous run; the six bounds of every group follow the histograms.
    const uint32_t n_groups = gridDim.x;
    uint4 *dst = reinterpret_cast<uint4 *>(partials);
    const uint4 *src = reinterpret_cast<const uint4 *>(bins);
#pragma unroll
    for (uint32_t i = threadIdx.x; i < kHistWords / 4; i += kHistBlock) {
        const uint4 v = src[i];
        __builtin_nontemporal_store((sum_u32x4){v.x, v.y, v.z, v.w},
                                    reinterpret_cast<sum_u32x4 *>(dst) + ((size_t)(i >> 4) * n_groups + blockIdx.x) * 16 + (i & 15));
    }
    if (threadIdx.x < 3) {
        uint32_t *bounds = partials + (size_t)n_groups * kHistWords + (size_t)blockIdx.x * 8;
        bounds[2 * threadIdx.x] = s_min[threadIdx.x];
        bounds[2 * threadIdx.x + 1] = s_max[threadIdx.x];
    }
}

// Second launch: bin b = sum over the groups' partials (16-bit pairs widened to u32), bounds = min / max over the groups.
// 256 workgroups x 1024 lanes: workgroup w owns 16 uint4 columns (= 128 bins); lane (c, s) adds column c of the partials
// s, s+64, ... (all of a 4K frame's 8 MiB in flight at once: with 256 lanes per workgroup and 8 loads per lane the launch was
// latency bound, 5.9 us at 128 groups and +2 us per 64 more); the four slices of a wave meet by shuffles, the 16 waves in LDS.
// `accumulate`: add to what hist / minmax already hold (frames of more than kMaxGroupsPerLaunch x 65535 samples take several
// launches).
constexpr int kReduceBlock = 1024;
constexpr uint32_t kReduceGroups = kHistWords / 4 / 16;

__global__ __launch_bounds__(kReduceBlock) void colordetect_reduce_kernel(const uint32_t *partials_all, uint32_t n_groups, uint32_t *hist_all,
                                                                          uint32_t *minmax_all, int accumulate)
{
    // blockIdx.y = frame (1 for the single-frame entry point): its partials, its 32768 + 8 output words
    const uint32_t *partials = partials_all + (size_t)blockIdx.y * n_groups * kHistWords;
    const uint32_t *bounds_in = partials_all + (size_t)gridDim.y * n_groups * kHistWords + (size_t)blockIdx.y * n_groups * 8;
    uint32_t *hist = hist_all + (size_t)blockIdx.y * (kHistBins + 8);
    uint32_t *minmax = minmax_all + (size_t)blockIdx.y * (kHistBins + 8);
    __shared__ uint32_t part[16][16 * 8];
    __shared__ uint32_t mm[32][8];
    const uint32_t c = threadIdx.x & 15, s = threadIdx.x >> 4; // 64 slices
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t g0 = s; g0 < n_groups; g0 += 64 * 4) { // four loads in flight per lane
        sum_u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t g = min(g0 + 64u * j, n_groups - 1);
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const sum_u32x4 *>(partials) + ((size_t)blockIdx.x * n_groups + g) * 16 + c);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (g0 + 64u * j >= n_groups) v[j] = (sum_u32x4){0, 0, 0, 0};
            acc[0] += v[j].x & 0xffffu; acc[1] += v[j].x >> 16;
            acc[2] += v[j].y & 0xffffu; acc[3] += v[j].y >> 16;
            acc[4] += v[j].z & 0xffffu; acc[5] += v[j].z >> 16;
            acc[6] += v[j].w & 0xffffu; acc[7] += v[j].w >> 16;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { // lanes c, c+16, c+32, c+48 of a wave hold the same column
        acc[k] += (uint32_t)__shfl_xor((int)acc[k], 16);
        acc[k] += (uint32_t)__shfl_xor((int)acc[k], 32);
    }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            part[threadIdx.x >> 6][c * 8 + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 16; k++)
            sum += part[k][threadIdx.x];
        uint32_t *out = hist + (size_t)blockIdx.x * 128 + threadIdx.x;
        *out = accumulate ? *out + sum : sum;
    }
    if (blockIdx.x == 0) { // bounds: value i of every group's six (min r, max r, min g, max g, min b, max b)
        const uint32_t i = threadIdx.x & 7, slice = (threadIdx.x >> 3) & 31;
        uint32_t m = (i & 1) ? 0u : 255u;
        if (i < 6 && threadIdx.x < 256)
            for (uint32_t g = slice; g < n_groups; g += 32) {
                const uint32_t v = bounds_in[(size_t)g * 8 + i];
                m = (i & 1) ? max(m, v) : min(m, v);
            }
        if (threadIdx.x < 256)
            mm[slice][i] = m;
        __syncthreads();
        if (threadIdx.x < 6) {
            uint32_t r = (i & 1) ? 0u : 255u;
            for (int k = 0; k < 32; k++)
                r = (i & 1) ? max(r, mm[k][i]) : min(r, mm[k][i]);
            if (accumulate)
                r = (i & 1) ? max(r, minmax[i]) : min(r, minmax[i]);
            minmax[i] = r;
        }
    }
}


// ---- round 3: the 4-byte formats on 16-byte aligned planes -------------------------------------------------------------------
// Every 64-byte line of the frame holds a sample at quality <= 10 (a sample every 40 bytes at most), so the kernel reads the
// WHOLE plane as a coalesced stream of 16-byte loads per lane (the blockhash shape, ~6 TB/s) and picks the samples out of the
// registers: pixel P is sample P / quality iff P % quality == 0.  A lane's unit is 4 pixels; remainder of its first pixel is
// kept incrementally (no division in the loop).  blockIdx.y = frame: the batched entry point bins one frame from each of n
// streams in one launch with few groups per frame (a group may take up to 65535 samples), so the dense partials stay small
// against the frames (n = 16 at 4K: 32 MiB of partials beside 531 MB of pixels).
constexpr int kHistMaxFrames = 32;

struct HistPlanes {
    const uint8_t *plane[kHistMaxFrames];
};

// kHist4InFlight: 16-byte loads a lane issues before it bins
template <bool MULTI, int kHist4InFlight> // MULTI: quality < 4, a 4-pixel unit can hold more than one sample
__global__ __launch_bounds__(kHistBlock) void colordetect_hist4_kernel(
    HistPlanes planes, uint64_t unit_begin, uint64_t unit_end, uint64_t px_begin, uint64_t px_end, uint32_t units_per_group,
    uint32_t quality, uint32_t step_mod_q, HistLayout lay, uint32_t *partials)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist_lds[];
    uint32_t *bins = hist_lds;
    uint32_t *s_min = hist_lds + kHistWords, *s_max = hist_lds + kHistWords + 4;
    for (uint32_t i = threadIdx.x; i < kHistWords; i += kHistBlock)
        bins[i] = 0;
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 255; s_max[threadIdx.x] = 0; }
    __syncthreads();

    const sum_u32x4 *src = reinterpret_cast<const sum_u32x4 *>(planes.plane[blockIdx.y]);
    const uint64_t g_begin = unit_begin + (uint64_t)blockIdx.x * units_per_group;
    const uint64_t g_end = min(g_begin + units_per_group, unit_end);
    uint32_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    const uint32_t sh_r = 8 * lay.ir, sh_g = 8 * lay.ig, sh_b = 8 * lay.ib, sh_a = 8 * lay.ia;
    uint64_t u0 = g_begin + threadIdx.x;
    uint32_t rem = (uint32_t)((u0 * 4ull) % quality); // (first pixel of the lane's unit) % quality, then incrementally
    // one pixel: skip test of color_thief (alpha < 125 or near white), 5-5-5 bin, LDS atomic on the packed pair
    auto bin_px = [&](uint32_t v) {
        uint32_t r = (v >> sh_r) & 0xff, g = (v >> sh_g) & 0xff, b = (v >> sh_b) & 0xff;
        const uint32_t a = (v >> sh_a) & 0xff;
        if (a < 125 || (r > 250 && g > 250 && b > 250))
            return;
        r >>= 3; g >>= 3; b >>= 3;
        mn[0] = min(mn[0], r); mx[0] = max(mx[0], r);
        mn[1] = min(mn[1], g); mx[1] = max(mx[1], g);
        mn[2] = min(mn[2], b); mx[2] = max(mx[2], b);
        const uint32_t bin = (r << 10) | (g << 5) | b;
        atomicAdd(&bins[bin >> 1], 1u << ((bin & 1) * 16));
    };
    for (; u0 < g_end; u0 += (uint64_t)kHistBlock * kHist4InFlight) {
        sum_u32x4 v[kHist4InFlight];
#pragma unroll
        for (int j = 0; j < kHist4InFlight; j++) { // unconditional loads from clamped units (no per-load s_waitcnt)
            const uint64_t u = min(u0 + (uint64_t)j * kHistBlock, g_end - 1);
            v[j] = __builtin_nontemporal_load(src + u);
        }
#pragma unroll
        for (int j = 0; j < kHist4InFlight; j++) {
            const uint64_t u = u0 + (uint64_t)j * kHistBlock;
            const bool in_group = u < g_end;
            const uint32_t rj = (rem + (uint32_t)j * ((kHistBlock * 4u) % quality)) % quality;
            const uint64_t p0 = u * 4ull;
            const uint32_t px[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            if (MULTI) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t t = rj + k; // < quality + 3 <= 3 * quality + ... : a sample iff t is a multiple of quality
                    const bool is_sample = (t == 0) | (t == quality) | (t == 2 * quality) | (t == 3 * quality);
                    if (in_group && is_sample && p0 + k >= px_begin && p0 + k < px_end)
                        bin_px(px[k]);
                }
            } else {
                // quality >= 4: at most one sample per unit, the pixel k = (quality - rj) % quality when that is < 4
                const uint32_t k = rj == 0 ? 0u : quality - rj;
                if (in_group && k < 4 && p0 + k >= px_begin && p0 + k < px_end) {
                    const uint32_t sel = k == 0 ? px[0] : k == 1 ? px[1] : k == 2 ? px[2] : px[3];
                    bin_px(sel);
                }
            }
        }
        rem = (rem + step_mod_q) % quality;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[c] = min(mn[c], (uint32_t)__shfl_down((int)mn[c], off));
            mx[c] = max(mx[c], (uint32_t)__shfl_down((int)mx[c], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_min[c], mn[c]);
            atomicMax(&s_max[c], mx[c]);
        }
    }
    __syncthreads();
    // dense flush, same layout as colordetect_hist_kernel, one region per frame: [frame][column block][group][16 uint4], then
    // the bounds of every group of every frame
    const uint32_t n_groups = gridDim.x;
    uint32_t *mine = partials + (size_t)blockIdx.y * n_groups * kHistWords;
    const uint4 *lds = reinterpret_cast<const uint4 *>(bins);
#pragma unroll
    for (uint32_t i = threadIdx.x; i < kHistWords / 4; i += kHistBlock) {
        const uint4 w = lds[i];
        __builtin_nontemporal_store((sum_u32x4){w.x, w.y, w.z, w.w},
                                    reinterpret_cast<sum_u32x4 *>(mine) + ((size_t)(i >> 4) * n_groups + blockIdx.x) * 16 + (i & 15));
    }
    if (threadIdx.x < 3) {
        uint32_t *bounds = partials + (size_t)gridDim.y * n_groups * kHistWords + ((size_t)blockIdx.y * n_groups + blockIdx.x) * 8;
        bounds[2 * threadIdx.x] = s_min[threadIdx.x];
        bounds[2 * threadIdx.x + 1] = s_max[threadIdx.x];
    }
}

int colordetect_layout(int format, HistLayout *lay)
{
    switch (format) { // colordetect/imp.rs:274-281 -> color_thief::ColorFormat
    case MVFX_FORMAT_RGB:  *lay = {3, 0, 1, 2, -1}; return 0;
    case MVFX_FORMAT_RGBA: *lay = {4, 0, 1, 2, 3}; return 0;
    case MVFX_FORMAT_ARGB: *lay = {4, 1, 2, 3, 0}; return 0;
    case MVFX_FORMAT_BGR:  *lay = {3, 2, 1, 0, -1}; return 0;
    case MVFX_FORMAT_BGRA: *lay = {4, 2, 1, 0, 3}; return 0;
    default: return -1;
    }
}

// One or several frames (same geometry, stride and format).  hist_dev / minmax_dev of frame f: hist_dev + f * hist_stride_words,
// minmax_dev + f * hist_stride_words (the frames entry point lays them out as n x (32768 + 8) words; the single-frame one passes
// its two pointers and n = 1).
int colordetect_hist_impl(const mvfx_frame *frames, uint32_t n_frames, uint32_t quality, uint64_t first_sample, uint64_t n_samples,
                          uint32_t *hist_dev, uint32_t *minmax_dev, hipStream_t st)
{
    if (!frames || !hist_dev || !minmax_dev || n_frames == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: NULL argument");
    const mvfx_frame *frame = &frames[0];
    HistLayout lay;
    if (colordetect_layout(frame->format, &lay) != 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "colordetect: format %d is not RGB RGBA ARGB BGR BGRA (colordetect/imp.rs:214-221)", frame->format);
    if (quality < 1 || quality > 10) // color-thief asserts quality in 1..=10
        return fail(MVFX_ERR_REFERENCE_PANIC, "colordetect: quality %u is outside 1..=10; color_thief::get_palette asserts on it", quality);
    bool aligned16 = true;
    for (uint32_t f = 0; f < n_frames; f++) {
        if (int rc = check_packed_frame(&frames[f], "colordetect"); rc != MVFX_OK) return rc;
        if (frames[f].width != frame->width || frames[f].height != frame->height || frames[f].stride != frame->stride ||
            frames[f].format != frame->format)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: the frames of one launch must share size, stride and format");
        aligned16 = aligned16 && (reinterpret_cast<uintptr_t>(frames[f].data) & 15) == 0;
    }
    if (n_frames > 1 && minmax_dev != hist_dev + kHistBins)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: batched output must be n x (32768 + 8) words");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    // every quality-th pixel of the flat plane, padding included (colordetect/imp.rs:69)
    const uint64_t pixel_count = (uint64_t)frame->stride * frame->height / (uint64_t)lay.bpp;
    const uint64_t total_samples = (pixel_count + quality - 1) / quality;
    if (first_sample > total_samples) first_sample = total_samples;
    if (n_samples > total_samples - first_sample) n_samples = total_samples - first_sample;

    static_assert(kReduceGroups * 128 == kHistBins, "the reduce kernel's workgroups own 128 bins each");
    if ((reinterpret_cast<uintptr_t>(hist_dev) & 15) != 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: the histogram buffer must be 16-byte aligned");
    int dev = 0;
    (void)hipGetDevice(&dev);
    constexpr size_t kHistLds = (kHistWords + 8) * sizeof(uint32_t);
    // per device, once per thread (not per frame): the CU count and the opt-in to > 64 KiB of dynamic LDS
    static thread_local int attr_device = -1, cus = 256;
    if (attr_device != dev) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        MVFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(colordetect_hist_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHistLds));
        const void *k4[] = {reinterpret_cast<const void *>(colordetect_hist4_kernel<false, 4>), reinterpret_cast<const void *>(colordetect_hist4_kernel<true, 4>)};
        for (const void *k : k4)
            MVFX_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHistLds));
        attr_device = dev;
    }
    int launches = 0;
    uint64_t byte_path_first = first_sample, byte_path_n = n_samples; // what is left for the per-sample kernel

    // ---- streaming path: 4-byte pixels, 16-byte aligned planes; covers the whole 4-pixel units of the sample range --------------
    // Measured on 4K RGBA (profiles/r3/colordetect_sweep.txt): one frame per launch, quality 10: per-sample kernel 10.7 + 5.1 us,
    // streaming kernel 13.0 + 6.0 us at its best (192 groups, 4 loads in flight; 96...256 groups x 4 / 8 / 12 in flight: 17.6-22.2 us
    // for the pair) -- a single frame is bound by the fixed cost of two launches and one memory round trip, which the sparse
    // 4-byte loads make as well; quality 1: 25.8 against 27.3 us.  16 frames per launch: 113.5 us = 7.1 us per frame = 0.58 of the
    // HBM peak (hist 95.7 us = 5.55 TB/s of frame reads, reduce 17.7 us), 16...64 groups per frame within 6 % of each other.
    const bool streaming = lay.bpp == 4 && aligned16 && n_samples > 0 && pixel_count >= 4 && (n_frames > 1 || quality < 4);
    if (streaming) {
        const uint64_t px_begin = first_sample * quality, px_end = (first_sample + n_samples - 1) * quality + 1; // sample pixels in [begin, end)
        const uint64_t unit_begin = px_begin / 4, unit_end = std::min<uint64_t>((px_end + 3) / 4, pixel_count / 4);
        // samples beyond the last whole unit (a plane whose pixel count is not a multiple of 4) go to the per-sample kernel below
        const uint64_t covered_end_px = std::min<uint64_t>(px_end, unit_end * 4);
        const uint64_t covered_samples = covered_end_px > px_begin ? (covered_end_px - 1 - px_begin) / quality + 1 : 0;
        byte_path_first = first_sample + covered_samples;
        byte_path_n = n_samples - covered_samples;
        // 16-bit bins: a group of U consecutive units (4U pixels) holds at most ceil(4U / quality) + 1 samples <= 65535
        const uint64_t cap_units = std::max<uint64_t>((uint64_t)(kMaxSamplesPerGroup - 1) * quality / 4 - 1, 1);
        uint64_t done = unit_begin;
        while (done < unit_end) {
            const uint64_t units = unit_end - done;
            // single frame: 3/4 of a 1024-lane group per CU (each group pays a 64 KiB LDS clear + a 64 KiB partial the second launch
            // reads back); several frames: ~1 group per CU in total, few per frame, so the partials stay small beside the pixels
            // (16 x 4K per launch pair, 8 / 12 / 16 / 24 / 32 groups per frame: 113.0 / 113.0 / 104.7 / 122.5 / 115.0 us,
            // profiles/r3/colordetect_groups.txt; an earlier box: 16 / 24 / 32 / 48 / 64: 105.2 / 120.9 / 113.6 / 115.3 / 120.9)
            const uint64_t want_groups = n_frames == 1 ? std::max<uint64_t>((uint64_t)cus * 3 / 4, 1)
                                                       : std::max<uint64_t>((uint64_t)cus / n_frames, 8);
            uint64_t per_group = std::max<uint64_t>((units + want_groups - 1) / want_groups, (uint64_t)kHistBlock * 4);
            per_group = std::min<uint64_t>(per_group, cap_units);
            uint64_t groups = (units + per_group - 1) / per_group;
            uint64_t chunk_units = units;
            if (groups > kMaxGroupsPerLaunch) { groups = kMaxGroupsPerLaunch; chunk_units = groups * per_group; }
            void *partials = nullptr;
            if (int rc = stream_scratch(st, (size_t)n_frames * groups * kPartialWords * sizeof(uint32_t), &partials); rc != MVFX_OK) return rc;
            for (uint32_t f0 = 0; f0 < n_frames; f0 += kHistMaxFrames) { // <= 32 plane pointers travel in the kernel arguments
                const uint32_t nf = std::min<uint32_t>(n_frames - f0, kHistMaxFrames);
                HistPlanes planes;
                for (uint32_t f = 0; f < nf; f++) planes.plane[f] = static_cast<const uint8_t *>(frames[f0 + f].data);
                uint32_t *part_f = static_cast<uint32_t *>(partials) + (size_t)f0 * groups * kPartialWords;
                const int in_flight = 4; // 16-byte loads in flight per lane (8 and 12 measured: never faster)
                const uint32_t step_mod = (uint32_t)(((uint64_t)kHistBlock * in_flight * 4) % quality);
                if (quality < 4)
                    MVFX_LAUNCH((colordetect_hist4_kernel<true, 4>), dim3((uint32_t)groups, nf), dim3(kHistBlock), kHistLds, st, planes, done,
                                       done + chunk_units, px_begin, px_end, (uint32_t)per_group, quality, step_mod, lay, part_f);
                else
                    MVFX_LAUNCH((colordetect_hist4_kernel<false, 4>), dim3((uint32_t)groups, nf), dim3(kHistBlock), kHistLds, st, planes, done,
                                       done + chunk_units, px_begin, px_end, (uint32_t)per_group, quality, step_mod, lay, part_f);
                MVFX_HIP_TRY(hipGetLastError());
                MVFX_LAUNCH(colordetect_reduce_kernel, dim3(kReduceGroups, nf), dim3(kReduceBlock), 0, st, part_f, (uint32_t)groups,
                                   hist_dev + (size_t)f0 * (kHistBins + 8), minmax_dev + (size_t)f0 * (kHistBins + 8), launches > 0 ? 1 : 0);
                MVFX_HIP_TRY(hipGetLastError());
            }
            done += chunk_units;
            launches++;
        }
        if (byte_path_n == 0)
            return MVFX_OK;
    }

    // ---- per-sample path: 3-byte pixels, unaligned planes, and the samples of a trailing partial unit ---------------------------
    for (uint32_t f = 0; f < n_frames; f++) {
        uint64_t done = 0;
        int frame_launches = launches;
        do {
            const uint64_t chunk = std::min<uint64_t>(byte_path_n - done, (uint64_t)kMaxGroupsPerLaunch * kMaxSamplesPerGroup);
            // 3/4 of a 1024-thread group per CU: every group pays a 64 KiB LDS clear and a 64 KiB partial that the second launch
            // has to read back, so few groups with many samples each win (4K, hist + reduce: 128 groups 11.4 + 4.8 us at quality 10
            // and 27.5 + 5.0 at quality 1; 192: 10.7 + 5.1 / 22.0 + 5.3; 256: 11.1 + 6.3 / 18.0 + 6.3); each group < 65536 samples
            // (16-bit bins)
            const uint64_t want_groups = std::max<uint64_t>((uint64_t)cus * 3 / 4, 1);
            uint32_t per_group = (uint32_t)std::min<uint64_t>(kMaxSamplesPerGroup, std::max<uint64_t>((chunk + want_groups - 1) / want_groups, 1024));
            const uint32_t groups = (uint32_t)((chunk + per_group - 1) / per_group);
            void *partials = nullptr;
            if (int rc = stream_scratch(st, (size_t)std::max<uint32_t>(groups, 1) * kPartialWords * sizeof(uint32_t), &partials); rc != MVFX_OK) return rc;
            if (groups) {
                MVFX_LAUNCH(colordetect_hist_kernel, dim3(groups), dim3(kHistBlock), kHistLds, st,
                                   static_cast<const uint8_t *>(frames[f].data), byte_path_first + done, chunk, per_group, quality,
                                   lay, static_cast<uint32_t *>(partials));
                MVFX_HIP_TRY(hipGetLastError());
            }
            MVFX_LAUNCH(colordetect_reduce_kernel, dim3(kReduceGroups, 1), dim3(kReduceBlock), 0, st,
                               static_cast<const uint32_t *>(partials), groups, hist_dev + (size_t)f * (kHistBins + 8),
                               minmax_dev + (size_t)f * (kHistBins + 8), frame_launches > 0 ? 1 : 0);
            MVFX_HIP_TRY(hipGetLastError());
            done += chunk;
            frame_launches++;
        } while (done < byte_path_n);
    }
    return MVFX_OK;
}

// ------------------------------------------------------------------ videocompare / blockhash

constexpr int kSumBlock = 256;
constexpr int kSumMaxPads = 16;      // frames per launch (blockIdx.z); more pads take more launches
constexpr int kSumRowsPerLane = 4;   // loads a lane has in flight (one unrolled batch), then the tail

struct SumPads {
    const uint8_t *plane[kSumMaxPads];
    uint64_t stride[kSumMaxPads];
};

// r+g+b of one RGBA pixel held in a dword (byte 3 = alpha): v_and + v_sad_u8; a fully transparent pixel counts as
// white (765).  alpha == 0  <=>  the dword is < 2^24.
__device__ __forceinline__ uint32_t rgba_brightness(uint32_t px)
{
    return px < 0x01000000u ? 765u : __builtin_amdgcn_sad_u8(px & 0x00ffffffu, 0u, 0u);
}

// One workgroup = 256 lanes laid out as LX x RY (LX = 2^lx_log2 lanes along the row, RY rows) inside ONE of the 64
// blocks (blockIdx.y) of ONE pad (blockIdx.z).  UNIT is what a lane loads at a time: 16 bytes = 4 RGBA px (uint4),
// 12 bytes = 4 RGB px (three dwords), or one pixel through byte loads when the frame is not aligned for those.
// No division in the loop; kSumRowsPerLane independent loads are issued before the first add.
template <int BPP, bool VEC>
__global__ __launch_bounds__(kSumBlock) void blockhash_sums_kernel(SumPads pads, uint32_t bw, uint32_t bh,
                                                                   uint32_t row_begin, uint32_t row_end,
                                                                   uint32_t lx_log2, uint32_t *partials)
{
    const uint32_t block = blockIdx.y;            // 0..63
    const uint32_t bx = block & 7, by = block >> 3;
    const uint8_t *plane = pads.plane[blockIdx.z];
    const uint64_t stride = pads.stride[blockIdx.z];
    uint32_t y0 = by * bh, y1 = y0 + bh;
    if (y0 < row_begin) y0 = row_begin;
    if (y1 > row_end) y1 = row_end;
    const uint32_t lx = 1u << lx_log2, ry = kSumBlock >> lx_log2;
    const uint32_t tx = threadIdx.x & (lx - 1), ty = threadIdx.x >> lx_log2;
    const uint32_t step = gridDim.x * ry;
    uint32_t acc = 0;
    const uint8_t *base = plane + (uint64_t)bx * bw * BPP;
    uint32_t r = y0 + blockIdx.x * ry + ty;
    if constexpr (VEC && BPP == 4) {
        const uint32_t units = bw >> 2; // uint4 per block row
        for (; r < y1 && y1 - r > (kSumRowsPerLane - 1) * step; r += kSumRowsPerLane * step) {
            for (uint32_t u = tx; u < units; u += lx) {
                sum_u32x4 v[kSumRowsPerLane];
#pragma unroll
                for (int k = 0; k < kSumRowsPerLane; k++)
                    v[k] = __builtin_nontemporal_load(reinterpret_cast<const sum_u32x4 *>(base + (uint64_t)(r + k * step) * stride) + u);
#pragma unroll
                for (int k = 0; k < kSumRowsPerLane; k++)
                    acc += rgba_brightness(v[k].x) + rgba_brightness(v[k].y) + rgba_brightness(v[k].z) + rgba_brightness(v[k].w);
            }
        }
        for (; r < y1; r += step)
            for (uint32_t u = tx; u < units; u += lx) {
                const sum_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const sum_u32x4 *>(base + (uint64_t)r * stride) + u);
                acc += rgba_brightness(v.x) + rgba_brightness(v.y) + rgba_brightness(v.z) + rgba_brightness(v.w);
            }
    } else if constexpr (VEC && BPP == 3) {
        const uint32_t units = bw >> 2; // 12 bytes = 4 RGB px: every byte counts, three v_sad_u8
        for (; r < y1 && y1 - r > (kSumRowsPerLane - 1) * step; r += kSumRowsPerLane * step) {
            for (uint32_t u = tx; u < units; u += lx) {
                uint32_t v[kSumRowsPerLane][3];
#pragma unroll
                for (int k = 0; k < kSumRowsPerLane; k++) {
                    const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (uint64_t)(r + k * step) * stride) + 3 * u;
                    v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2];
                }
#pragma unroll
                for (int k = 0; k < kSumRowsPerLane; k++)
                    acc = __builtin_amdgcn_sad_u8(v[k][2], 0u, __builtin_amdgcn_sad_u8(v[k][1], 0u, __builtin_amdgcn_sad_u8(v[k][0], 0u, acc)));
            }
        }
        for (; r < y1; r += step)
            for (uint32_t u = tx; u < units; u += lx) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (uint64_t)r * stride) + 3 * u;
                acc = __builtin_amdgcn_sad_u8(p[2], 0u, __builtin_amdgcn_sad_u8(p[1], 0u, __builtin_amdgcn_sad_u8(p[0], 0u, acc)));
            }
    } else {
        for (; r < y1; r += step)
            for (uint32_t x = tx; x < bw; x += lx) {
                const uint8_t *p = base + (uint64_t)r * stride + (uint64_t)x * BPP;
                if constexpr (BPP == 4)
                    acc += p[3] == 0 ? 765u : (uint32_t)p[0] + p[1] + p[2];
                else
                    acc += (uint32_t)p[0] + p[1] + p[2];
            }
    }
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off);
    __shared__ uint32_t wave_sum[kSumBlock / 64];
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < kSumBlock / 64; i++) t += wave_sum[i];
        partials[(blockIdx.z * 64 + block) * gridDim.x + blockIdx.x] = t; // plain store, see blockhash_reduce_kernel
    }
}

// Second (tiny) launch: one wave per (pad, block) adds that block's per-workgroup partials and OVERWRITES sums[].
// Same-address atomics straight from the first kernel cost ~0.25 us each once a few hundred workgroups hit one of
// the 64 addresses (8K frame: 65 us with 135 atomics per address against 23.5 us with stores + this launch,
// tools/probe_blockhash.hip); this launch also replaces the memset the atomics needed.
__global__ __launch_bounds__(64) void blockhash_reduce_kernel(const uint32_t *partials, uint32_t per_block, uint32_t *sums)
{
    uint32_t a = 0;
    for (uint32_t i = threadIdx.x; i < per_block; i += 64) a += partials[blockIdx.x * per_block + i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (threadIdx.x == 0) sums[blockIdx.x] = a;
}

// ---- image_hasher's `blockhash_slow` (frames whose width or height is not a multiple of 8) ------------------------
// f32 block sums accumulated in RASTER ORDER, up to four `+=` per pixel (see oracle/videofx_oracle.c for the crate's
// statements incl. its `x + 1. % block_width` precedence).  Every pixel lands in exactly one block (right == left and
// bottom == top always), so the 64 blocks are 64 independent ordered chains: one wave per (block, pad).
//   UNIT weights (width > 8 and height > 8: fmod(1, block) == 1, weights exactly 0 / 1): a pixel adds its integer value
//   once (+0.0 three times: no-ops).  While the running total is certainly < 2^24 every f32 addition is exact and the
//   order cannot matter: the first kExactPixels pixels of a block (raster order) are summed in parallel; after that the
//   chain is followed literally, one v_add_f32 per pixel in raster order (lane values through v_readlane).
//   Other frames (a dimension <= 8): fractional weights, four literal additions per pixel, one lane per block.
struct SlowBounds {
    uint32_t xs[9], ys[9]; // block b covers columns [xs[b], xs[b+1]) / rows [ys[b], ys[b+1]): floor(x / block_width) in f32, host
    float mx, my;          // fmodf(1, block_width), fmodf(1, block_height)
};
constexpr uint32_t kExactPixels = (1u << 24) / 765u - 64u; // 21,867 pixels x 765 < 2^24

template <int BPP>
__device__ __forceinline__ uint32_t slow_pixel_value(const uint8_t *p)
{
    if constexpr (BPP == 4) return p[3] == 0 ? 765u : (uint32_t)p[0] + p[1] + p[2];
    else return (uint32_t)p[0] + p[1] + p[2];
}

template <int BPP>
__global__ __launch_bounds__(64) void blockhash_slow_unit_kernel(SumPads pads, SlowBounds b, uint32_t *sums)
{
    const uint32_t block = blockIdx.x, bx = block & 7, by = block >> 3, lane = threadIdx.x;
    const uint8_t *plane = pads.plane[blockIdx.y];
    const uint64_t stride = pads.stride[blockIdx.y];
    const uint32_t x0 = b.xs[bx], x1 = b.xs[bx + 1], y0 = b.ys[by], y1 = b.ys[by + 1];
    uint32_t consumed = 0, acc = 0; // exact phase: private integer partials, `consumed` is wave-uniform
    float fsum = 0.0f;
    bool exact = true;
    for (uint32_t y = y0; y < y1; y++) {
        const uint8_t *row = plane + (uint64_t)y * stride;
        for (uint32_t xc = x0; xc < x1; xc += 64) {
            const uint32_t x = xc + lane;
            const uint32_t v = x < x1 ? slow_pixel_value<BPP>(row + (uint64_t)x * BPP) : 0u;
            const uint32_t n = x1 - xc < 64u ? x1 - xc : 64u;
            if (exact && consumed + n <= kExactPixels) {
                acc += v;
                consumed += n;
                continue;
            }
            if (exact) { // leave the exact phase: total of the prefix, < 2^24, exactly representable
                exact = false;
                uint32_t t = acc;
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
                fsum = (float)t;
            }
            const float fv = (float)v; // lanes past the block's last column hold +0.0: adding it changes nothing
#pragma unroll
            for (int i = 0; i < 64; i++)
                fsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fv), i));
        }
    }
    if (exact) {
        uint32_t t = acc;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        fsum = (float)t;
    }
    if (lane == 0) sums[blockIdx.y * 64 + block] = __builtin_bit_cast(uint32_t, fsum);
}

// fractional weights (a dimension <= 8 pixels): literal transcription, thread = block
template <int BPP>
__global__ __launch_bounds__(64) void blockhash_slow_weighted_kernel(SumPads pads, SlowBounds b, uint32_t *sums)
{
    const uint32_t block = threadIdx.x, bx = block & 7, by = block >> 3;
    const uint8_t *plane = pads.plane[blockIdx.x];
    const uint64_t stride = pads.stride[blockIdx.x];
    float sum = 0.0f;
    for (uint32_t yi = b.ys[by]; yi < b.ys[by + 1]; yi++) {
        const float y_mod = (float)yi + b.my;
        const float weight_top = y_mod - truncf(y_mod), weight_bottom = 1.0f - weight_top;
        for (uint32_t xi = b.xs[bx]; xi < b.xs[bx + 1]; xi++) {
            const float x_mod = (float)xi + b.mx;
            const float weight_left = x_mod - truncf(x_mod), weight_right = 1.0f - weight_left;
            const float px_sum = (float)slow_pixel_value<BPP>(plane + (uint64_t)yi * stride + (uint64_t)xi * BPP);
            sum += px_sum * weight_left * weight_top;
            sum += px_sum * weight_left * weight_bottom;
            sum += px_sum * weight_right * weight_top;
            sum += px_sum * weight_right * weight_bottom;
        }
    }
    sums[blockIdx.x * 64 + block] = __builtin_bit_cast(uint32_t, sum);
}

// Host side of the slow path: block boundaries with the crate's f32 arithmetic (`(x as f32 / block_width).floor()`),
// then one wave per (block, pad).  Not shardable by rows: a block sum is ONE ordered f32 chain over all of its rows.
int blockhash_slow_impl(const mvfx_frame *frames, uint32_t n_pads, uint32_t *sums_dev, hipStream_t st)
{
    const mvfx_frame *f0 = &frames[0];
    const uint32_t w = f0->width, h = f0->height;
    SlowBounds b{};
    const float block_width = (float)w / 8.0f, block_height = (float)h / 8.0f;
    b.mx = std::fmod(1.0f, block_width);
    b.my = std::fmod(1.0f, block_height);
    auto bounds = [](uint32_t n, float block, uint32_t out[9]) -> bool {
        uint32_t prev = 0;
        out[0] = 0;
        for (uint32_t i = 0; i < n; i++) {
            const float q = std::floor((float)i / block);
            if (!(q >= 0.0f && q <= 7.0f)) return false; // the crate would index out of bounds
            const uint32_t blk = (uint32_t)q;
            if (blk < prev) return false;
            for (; prev < blk; prev++) out[prev + 1] = i;
        }
        for (; prev < 8; prev++) out[prev + 1] = n;
        return true;
    };
    if (!bounds(w, block_width, b.xs) || !bounds(h, block_height, b.ys))
        return fail(MVFX_ERR_REFERENCE_PANIC, "blockhash: block index out of range for %ux%u (image_hasher would panic)", w, h);
    const bool unit = b.mx == 1.0f && b.my == 1.0f;
    const int bpp = f0->format == MVFX_FORMAT_RGBA ? 4 : 3;
    for (uint32_t first = 0; first < n_pads; first += kSumMaxPads) {
        const uint32_t n = std::min<uint32_t>(kSumMaxPads, n_pads - first);
        SumPads pads{};
        for (uint32_t p = 0; p < n; p++) {
            pads.plane[p] = static_cast<const uint8_t *>(frames[first + p].data);
            pads.stride[p] = frames[first + p].stride;
        }
        uint32_t *out = sums_dev + (size_t)first * 64;
        if (unit) {
            if (bpp == 4) MVFX_LAUNCH(blockhash_slow_unit_kernel<4>, dim3(64, n), dim3(64), 0, st, pads, b, out);
            else MVFX_LAUNCH(blockhash_slow_unit_kernel<3>, dim3(64, n), dim3(64), 0, st, pads, b, out);
        } else {
            if (bpp == 4) MVFX_LAUNCH(blockhash_slow_weighted_kernel<4>, dim3(n), dim3(64), 0, st, pads, b, out);
            else MVFX_LAUNCH(blockhash_slow_weighted_kernel<3>, dim3(n), dim3(64), 0, st, pads, b, out);
        }
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}

// Block sums of rows [row_begin,row_end) of n_pads frames of one size and format -> sums_dev[n_pads][64]
int blockhash_sums_impl(const mvfx_frame *frames, uint32_t n_pads, uint32_t row_begin, uint32_t row_end,
                        uint32_t *sums_dev, hipStream_t st)
{
    if (!frames || !sums_dev || n_pads == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: NULL argument");
    const mvfx_frame *f0 = &frames[0];
    for (uint32_t p = 0; p < n_pads; p++) {
        const mvfx_frame *frame = &frames[p];
        if (frame->format != MVFX_FORMAT_RGB && frame->format != MVFX_FORMAT_RGBA)
            return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "videocompare: format %d is not RGB / RGBA (videocompare/imp.rs:160-162)", frame->format);
        if (int rc = check_packed_frame(frame, "videocompare"); rc != MVFX_OK) return rc;
        if (frame->width != f0->width || frame->height != f0->height)
            return fail(MVFX_ERR_NOT_NEGOTIATED, "Video streams do not have the same sizes (videocompare/imp.rs:337-346)");
        if (frame->format != f0->format)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: the frames of one multi-pad call must share one format");
    }
    if (f0->width == 0 || f0->height == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: empty frame");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    if (row_end > f0->height) row_end = f0->height;
    if (f0->width % 8 != 0 || f0->height % 8 != 0) {
        // image_hasher's f32 path: a block sum is one ordered chain over all of the block's rows -> whole frames only
        if (row_begin != 0 || row_end != f0->height)
            return fail(MVFX_ERR_INVALID_ARGUMENT,
                        "blockhash: %ux%u is not a multiple of 8 in both dimensions: the f32 block sums of that path are "
                        "order-dependent chains and cannot be split into row bands", f0->width, f0->height);
        return blockhash_slow_impl(frames, n_pads, sums_dev, st);
    }
    if (row_begin >= row_end) {
        MVFX_HIP_TRY(hipMemsetAsync(sums_dev, 0, (size_t)n_pads * 64 * sizeof(uint32_t), st));
        return MVFX_OK;
    }
    const uint32_t bw = f0->width / 8, bh = f0->height / 8;
    const int bpp = f0->format == MVFX_FORMAT_RGBA ? 4 : 3;
    for (uint32_t first = 0; first < n_pads; first += kSumMaxPads) {
        const uint32_t n = std::min<uint32_t>(kSumMaxPads, n_pads - first);
        SumPads pads{};
        bool vec = (bw & 3) == 0;
        for (uint32_t p = 0; p < n; p++) {
            pads.plane[p] = static_cast<const uint8_t *>(frames[first + p].data);
            pads.stride[p] = frames[first + p].stride;
            // 16-byte groups need 16-byte aligned block rows (RGBA); 12-byte groups need dword alignment (RGB)
            const uintptr_t bits = reinterpret_cast<uintptr_t>(pads.plane[p]) | pads.stride[p] | (uintptr_t)((uint64_t)bw * bpp);
            vec = vec && (bits & (bpp == 4 ? 15 : 3)) == 0;
        }
        const uint32_t units = vec ? bw >> 2 : bw;
        uint32_t lx_log2 = 0;
        while ((1u << lx_log2) < units && lx_log2 < 8) lx_log2++;
        const uint32_t ry = kSumBlock >> lx_log2;
        const uint32_t passes = (units + (1u << lx_log2) - 1) >> lx_log2; // loads per lane per row
        // each lane: kSumRowsPerLane rows of `passes` loads (passes > 1 only for frames wider than 8192 px)
        (void)passes;
        uint32_t chunks = (bh + ry * kSumRowsPerLane - 1) / (ry * kSumRowsPerLane);
        if (chunks < 1) chunks = 1;
        if (chunks > 1024) chunks = 1024;
        const dim3 grid(chunks, 64, n);
        void *scratch = nullptr; // per-workgroup partials [pad][block][chunk], owned by the calling thread
        if (int rc = host_scratch((size_t)n * 64 * chunks * sizeof(uint32_t), 2, &scratch); rc != MVFX_OK) return rc;
        uint32_t *out = static_cast<uint32_t *>(scratch);
        if (bpp == 4 && vec)
            MVFX_LAUNCH((blockhash_sums_kernel<4, true>), grid, dim3(kSumBlock), 0, st, pads, bw, bh, row_begin, row_end, lx_log2, out);
        else if (bpp == 4)
            MVFX_LAUNCH((blockhash_sums_kernel<4, false>), grid, dim3(kSumBlock), 0, st, pads, bw, bh, row_begin, row_end, lx_log2, out);
        else if (vec)
            MVFX_LAUNCH((blockhash_sums_kernel<3, true>), grid, dim3(kSumBlock), 0, st, pads, bw, bh, row_begin, row_end, lx_log2, out);
        else
            MVFX_LAUNCH((blockhash_sums_kernel<3, false>), grid, dim3(kSumBlock), 0, st, pads, bw, bh, row_begin, row_end, lx_log2, out);
        MVFX_LAUNCH(blockhash_reduce_kernel, dim3(64 * n), dim3(64), 0, st, out, chunks, sums_dev + (size_t)first * 64);
        MVFX_HIP_TRY(hipGetLastError());
    }
    return MVFX_OK;
}

// bits from the 64 sums (gen_hash!): 4 groups of 16 blocks, upper median, `> median` or equal-and-bright.
// Multiples of 8: u32 sums, equality exact.  Other sizes: the words are f32 bit patterns, |v - median| < 0.001
// (FLOAT_EQ_MARGIN) and the brightness bound 765.0 * (block_width * block_height) / 2.0 in f32.
uint64_t blockhash_bits(const uint32_t sums[64], uint32_t width, uint32_t height)
{
    uint64_t hash = 0;
    if (width % 8 != 0 || height % 8 != 0) {
        float blocks[64];
        std::memcpy(blocks, sums, sizeof(blocks));
        const float block_area = ((float)width / 8.0f) * ((float)height / 8.0f);
        const float cmp_factor = 765.0f * block_area / 2.0f;
        for (int band = 0; band < 4; band++) {
            float sorted[16];
            std::memcpy(sorted, blocks + 16 * band, sizeof(sorted));
            std::nth_element(sorted, sorted + 8, sorted + 16);
            const float median = sorted[8];
            for (int i = 0; i < 16; i++) {
                const float v = blocks[16 * band + i];
                if (v > median || (std::fabs(v - median) < 0.001f && median > cmp_factor))
                    hash |= 1ull << (16 * band + i);
            }
        }
        return hash;
    }
    const uint64_t half_block_value = (uint64_t)765 * (width / 8) * (height / 8) / 2;
    for (int band = 0; band < 4; band++) {
        uint32_t sorted[16];
        std::memcpy(sorted, sums + 16 * band, sizeof(sorted));
        std::nth_element(sorted, sorted + 8, sorted + 16);
        const uint32_t median = sorted[8];
        for (int i = 0; i < 16; i++) {
            const uint32_t v = sums[16 * band + i];
            if (v > median || (v == median && (uint64_t)median > half_block_value))
                hash |= 1ull << (16 * band + i);
        }
    }
    return hash;
}

// ------------------------------------------------------------------ roundedcorners

// The four plane copies of the A420 compose in ONE launch (grid z = plane): four back-to-back launches of
// 2-8 MB each were launch-bound (20 us for 33 MB).  A plane whose rows are contiguous in both buffers is
// described as a single long row.
struct PlaneCopy {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t row_bytes, src_stride, dst_stride;
    uint32_t rows;
};
struct PlaneCopies {
    PlaneCopy p[4];
};

// non-temporal: the planes stream through once; alone the kernel runs the same 11 us per 4K compose, beside colordetect on another
// stream the pair gains 2.7 % (44.0 k -> 45.2 k frames/s: the histogram partials keep the L2)
#ifndef MVFX_COPY_PLANES_NT
#define MVFX_COPY_PLANES_NT 2
#endif
__global__ __launch_bounds__(256) void copy_planes_kernel(PlaneCopies pc)
{
    const PlaneCopy c = pc.p[blockIdx.z];
    for (uint32_t y = blockIdx.y; y < c.rows; y += gridDim.y) {
        const uint8_t *s = c.src + (uint64_t)y * c.src_stride;
        uint8_t *d = c.dst + (uint64_t)y * c.dst_stride;
        const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
        const uint64_t step = (uint64_t)gridDim.x * 256;
        if (vec) {
            const uint64_t n16 = c.row_bytes >> 4;
            for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += step) {
#if MVFX_COPY_PLANES_NT == 2 // non-temporal load + the write-through store of csrc/device_store.hpp: 4K I420 -> A420 compose 89.1 k -> 91.4-92.2 k fps
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(s) + i);
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(reinterpret_cast<u32x4_t *>(d) + i), "v"(t) : "memory");
#elif MVFX_COPY_PLANES_NT // (rounds 3-5: non-temporal load + store)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(s) + i), reinterpret_cast<u32x4_t *>(d) + i);
#else
                reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
#endif
            }
            for (uint64_t i = (n16 << 4) + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < c.row_bytes; i += step)
                d[i] = s[i];
        } else {
            for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < c.row_bytes; i += step)
                d[i] = s[i];
        }
    }
}

int launch_copy_planes(PlaneCopies pc, int n, hipStream_t st)
{
    uint64_t max_row = 0;
    uint32_t max_rows = 0;
    for (int i = 0; i < 4; i++) {
        PlaneCopy &c = pc.p[i];
        if (i >= n) { c.rows = 0; c.row_bytes = 0; continue; }
        if (c.rows > 1 && c.src_stride == c.row_bytes && c.dst_stride == c.row_bytes) { // contiguous: one long row
            c.row_bytes *= c.rows;
            c.rows = 1;
        }
        max_row = std::max(max_row, c.row_bytes);
        max_rows = std::max(max_rows, c.rows);
    }
    if (max_rows == 0 || max_row == 0) return MVFX_OK;
    uint64_t bx = (max_row / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 65535u) bx = 65535u; // grid-stride covers the rest
    MVFX_LAUNCH(copy_planes_kernel, dim3((uint32_t)bx, max_rows < 65535u ? max_rows : 65535u, n), dim3(256), 0, st, pc);
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_colordetect_histogram(const mvfx_frame *frame, uint32_t quality, uint64_t first_sample, uint64_t n_samples,
                               uint32_t *hist_device, uint32_t *minmax_device, mvfx_stream stream)
{
    return colordetect_hist_impl(frame, 1, quality, first_sample, n_samples, hist_device, minmax_device, as_stream(stream));
}

int mvfx_colordetect_histogram_frames(const mvfx_frame *frames, uint32_t n_frames, uint32_t quality, uint32_t *hist_device,
                                      mvfx_stream stream)
{
    if (!hist_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: NULL output");
    return colordetect_hist_impl(frames, n_frames, quality, 0, ~0ull, hist_device, hist_device + kHistBins, as_stream(stream));
}

int mvfx_mmcq_palette_from_histogram(const uint32_t *hist_host, const uint32_t minmax[6], uint32_t max_colors,
                                     uint32_t *palette_out, uint32_t *n_out)
{
    if (!hist_host || !minmax || !palette_out || !n_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "mmcq: NULL argument");
    if (max_colors < 2 || max_colors > 255) // color-thief: assert!(max_colors > 1), u8 argument
        return fail(MVFX_ERR_REFERENCE_PANIC, "colordetect: max-colors %u is outside 2..=255", max_colors);
    const std::vector<Rgb8> pal = mmcq_palette(hist_host, minmax, max_colors);
    if (pal.empty())
        return fail(MVFX_ERR_DEVICE, "colordetect: palette extraction failed (color_thief error -> FlowError::Error, colordetect/imp.rs:74)");
    for (size_t i = 0; i < pal.size(); i++)
        palette_out[i] = ((uint32_t)pal[i].r << 16) | ((uint32_t)pal[i].g << 8) | pal[i].b; // colordetect/imp.rs:95-99
    *n_out = (uint32_t)pal.size();
    return MVFX_OK;
}

const char *mvfx_css_color_similar(uint8_t r, uint8_t g, uint8_t b) { return css_color_similar(r, g, b); }

static int palette_common(const mvfx_frame *dev_frame, uint32_t quality, uint32_t max_colors, uint32_t *palette_out,
                          uint32_t *n_out, hipStream_t st)
{
    if (!palette_out || !n_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: NULL output");
    if (max_colors < 2 || max_colors > 255)
        return fail(MVFX_ERR_REFERENCE_PANIC, "colordetect: max-colors %u is outside 2..=255", max_colors);
    void *scratch = nullptr;
    if (int rc = host_scratch((kHistBins + 8) * sizeof(uint32_t), 3, &scratch); rc != MVFX_OK) return rc;
    uint32_t *hist_dev = static_cast<uint32_t *>(scratch), *mm_dev = hist_dev + kHistBins;
    if (int rc = colordetect_hist_impl(dev_frame, 1, quality, 0, ~0ull, hist_dev, mm_dev, st); rc != MVFX_OK) return rc;
    // the 128 KiB histogram into page-locked memory (one block per thread, kept): a D2H copy to pageable memory is staged
    static thread_local uint32_t *pinned = nullptr;
    if (!pinned) {
        void *q = nullptr;
        if (hipHostMalloc(&q, (kHistBins + 8) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) pinned = static_cast<uint32_t *>(q);
        else (void)hipGetLastError();
    }
    std::vector<uint32_t> pageable;
    uint32_t *host = pinned;
    if (!host) {
        pageable.resize(kHistBins + 8);
        host = pageable.data();
    }
    MVFX_HIP_TRY(hipMemcpyAsync(host, hist_dev, (kHistBins + 6) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return mvfx_mmcq_palette_from_histogram(host, host + kHistBins, max_colors, palette_out, n_out);
}

int mvfx_colordetect_palette(const mvfx_frame *frame, uint32_t quality, uint32_t max_colors, uint32_t *palette_out,
                             uint32_t *n_out, mvfx_stream stream)
{
    return palette_common(frame, quality, max_colors, palette_out, n_out, as_stream(stream));
}

int mvfx_colordetect_palette_host(const mvfx_frame *frame, uint32_t quality, uint32_t max_colors,
                                  uint32_t *palette_out, uint32_t *n_out)
{
    if (!frame)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "colordetect: NULL frame");
    if (int rc = check_packed_frame(frame, "colordetect"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t bytes = (size_t)frame->stride * frame->height;
    void *dev = nullptr;
    if (int rc = host_scratch(bytes ? bytes : 16, 0, &dev); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    if (bytes)
        MVFX_HIP_TRY(hipMemcpyAsync(dev, frame->data, bytes, hipMemcpyHostToDevice, st));
    mvfx_frame d = *frame;
    d.data = dev;
    return palette_common(&d, quality, max_colors, palette_out, n_out, st);
}

int mvfx_blockhash_sums(const mvfx_frame *frame, uint32_t row_begin, uint32_t row_end, uint32_t *sums_device,
                        mvfx_stream stream)
{
    return blockhash_sums_impl(frame, 1, row_begin, row_end, sums_device, as_stream(stream));
}

} // extern "C"

namespace mvfx {
// views of whole frames whose rows outside the band are never touched (also called by comm_rccl.hip)
int blockhash_bands_impl(const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height, uint32_t band_first_row,
                                uint32_t *sums_device, hipStream_t st)
{
    if (!bands || !sums_device || n_pads == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: NULL argument");
    std::vector<mvfx_frame> whole(bands, bands + n_pads);
    bool empty = false;
    for (uint32_t p = 0; p < n_pads; p++) {
        if ((uint64_t)band_first_row + bands[p].height > full_height)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: band rows %u..%u exceed the frame height %u", band_first_row,
                        band_first_row + bands[p].height, full_height);
        if (bands[p].height != bands[0].height)
            return fail(MVFX_ERR_NOT_NEGOTIATED, "Video streams do not have the same sizes (videocompare/imp.rs:337-346)");
        empty = empty || bands[p].height == 0 || bands[p].width == 0;
        whole[p].height = full_height;
        whole[p].data = static_cast<uint8_t *>(bands[p].data) - (ptrdiff_t)((uint64_t)band_first_row * bands[p].stride);
    }
    if (empty) {
        if (int rc = require_device(); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipMemsetAsync(sums_device, 0, (size_t)n_pads * 64 * sizeof(uint32_t), st));
        return MVFX_OK;
    }
    return blockhash_sums_impl(whole.data(), n_pads, band_first_row, band_first_row + bands[0].height, sums_device, st);
}
} // namespace mvfx

extern "C" {

int mvfx_blockhash_sums_band(const mvfx_frame *band, uint32_t full_height, uint32_t band_first_row,
                             uint32_t *sums_device, mvfx_stream stream)
{
    return blockhash_bands_impl(band, 1, full_height, band_first_row, sums_device, as_stream(stream));
}

int mvfx_blockhash_sums_pads(const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height, uint32_t band_first_row,
                             uint32_t *sums_device, mvfx_stream stream)
{
    return blockhash_bands_impl(bands, n_pads, full_height, band_first_row, sums_device, as_stream(stream));
}

int mvfx_blockhash_bits(const uint32_t sums_host[64], uint32_t width, uint32_t height, uint64_t *hash_out)
{
    if (!sums_host || !hash_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash_bits: NULL argument");
    if (width == 0 || height == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash_bits: empty frame");
    *hash_out = blockhash_bits(sums_host, width, height);
    return MVFX_OK;
}

uint32_t mvfx_hash_distance(uint64_t a, uint64_t b) { return (uint32_t)__builtin_popcountll(a ^ b); }

// 2 x 64 block sums land in page-locked host memory straight from the reduce kernel's stores (hipHostMalloc memory is device-accessible
// and coherent): no D2H copy packet behind the kernel, and nothing staged through pageable memory.  One block per thread, kept.
static uint32_t *pinned_sums()
{
    static thread_local uint32_t *p = nullptr;
    if (!p) {
        void *q = nullptr;
        if (hipHostMalloc(&q, 128 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        p = static_cast<uint32_t *>(q);
    }
    return p;
}

static int blockhash_common(const mvfx_frame *dev_frame, uint64_t *hash_out, hipStream_t st)
{
    if (!hash_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: NULL output");
    void *scratch = nullptr;
    if (int rc = host_scratch(64 * sizeof(uint32_t), 3, &scratch); rc != MVFX_OK) return rc;
    uint32_t *sums_dev = static_cast<uint32_t *>(scratch);
    uint32_t sums[64];
    if (uint32_t *host = pinned_sums()) {
        if (int rc = blockhash_sums_impl(dev_frame, 1, 0, dev_frame ? dev_frame->height : 0, host, st); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipStreamSynchronize(st));
        memcpy(sums, host, sizeof(sums));
    } else {
        if (int rc = blockhash_sums_impl(dev_frame, 1, 0, dev_frame ? dev_frame->height : 0, sums_dev, st); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipMemcpyAsync(sums, sums_dev, sizeof(sums), hipMemcpyDeviceToHost, st));
        MVFX_HIP_TRY(hipStreamSynchronize(st));
    }
    *hash_out = blockhash_bits(sums, dev_frame->width, dev_frame->height);
    return MVFX_OK;
}

int mvfx_blockhash(const mvfx_frame *frame, uint64_t *hash_out, mvfx_stream stream)
{
    return blockhash_common(frame, hash_out, as_stream(stream));
}

int mvfx_blockhash_host(const mvfx_frame *frame, uint64_t *hash_out)
{
    if (!frame)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "blockhash: NULL frame");
    if (int rc = check_packed_frame(frame, "videocompare"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t bytes = (size_t)frame->stride * frame->height;
    void *dev = nullptr;
    if (int rc = host_scratch(bytes ? bytes : 16, 0, &dev); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    if (bytes)
        MVFX_HIP_TRY(hipMemcpyAsync(dev, frame->data, bytes, hipMemcpyHostToDevice, st));
    mvfx_frame d = *frame;
    d.data = dev;
    return blockhash_common(&d, hash_out, st);
}

int mvfx_videocompare_distance(const mvfx_frame *reference_frame, const mvfx_frame *other_frame, double *distance_out,
                               mvfx_stream stream)
{
    if (!reference_frame || !other_frame || !distance_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: NULL argument");
    if (reference_frame->width != other_frame->width || reference_frame->height != other_frame->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "Video streams do not have the same sizes (videocompare/imp.rs:337-346)");
    // both pads in ONE launch, one D2H of 2 x 64 sums, one synchronisation
    void *scratch = nullptr;
    if (int rc = host_scratch(2 * 64 * sizeof(uint32_t), 3, &scratch); rc != MVFX_OK) return rc;
    uint32_t *sums_dev = static_cast<uint32_t *>(scratch);
    hipStream_t st = as_stream(stream);
    uint32_t sums[128];
    uint32_t *const host = pinned_sums();
    if (host) sums_dev = host; // the reduce kernel stores into page-locked host memory: no copy behind it
    if (reference_frame->format == other_frame->format) {
        const mvfx_frame pair[2] = {*reference_frame, *other_frame};
        if (int rc = blockhash_sums_impl(pair, 2, 0, pair[0].height, sums_dev, st); rc != MVFX_OK) return rc;
    } else { // RGB against RGBA: one launch per pad
        if (int rc = blockhash_sums_impl(reference_frame, 1, 0, reference_frame->height, sums_dev, st); rc != MVFX_OK) return rc;
        if (int rc = blockhash_sums_impl(other_frame, 1, 0, other_frame->height, sums_dev + 64, st); rc != MVFX_OK) return rc;
    }
    if (!host) MVFX_HIP_TRY(hipMemcpyAsync(sums, sums_dev, sizeof(sums), hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    if (host) memcpy(sums, host, sizeof(sums));
    const uint64_t a = blockhash_bits(sums, reference_frame->width, reference_frame->height);
    const uint64_t b = blockhash_bits(sums + 64, other_frame->width, other_frame->height);
    *distance_out = (double)mvfx_hash_distance(a, b); // hashed_image.rs:70 `left.dist(right) as f64`
    return MVFX_OK;
}

int mvfx_roundedcorners_mask_host(uint8_t *mask_host, uint32_t width, uint32_t height, uint32_t stride,
                                  uint32_t border_radius_px)
{
    if (!mask_host || stride < width || width == 0 || height == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: bad mask geometry %ux%u stride %u", width, height, stride);
    return cairo_render_rounded_mask(mask_host, width, height, stride, border_radius_px);
}

int mvfx_roundedcorners_mask(uint8_t *mask_device, uint32_t width, uint32_t height, uint32_t stride,
                             uint32_t border_radius_px, mvfx_stream stream)
{
    if (!mask_device || stride < width || width == 0 || height == 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: bad mask geometry %ux%u stride %u", width, height, stride);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    // once per caps / border-radius change (border/imp.rs:491-519): render with cairo, move to HBM
    const size_t bytes = (size_t)stride * ((height + 1) & ~1u);
    std::vector<uint8_t> host(bytes);
    if (int rc = cairo_render_rounded_mask(host.data(), width, height, stride, border_radius_px); rc != MVFX_OK) return rc;
    hipStream_t st = as_stream(stream);
    MVFX_HIP_TRY(hipMemcpyAsync(mask_device, host.data(), bytes, hipMemcpyHostToDevice, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st)); // `host` dies with this call
    return MVFX_OK;
}

const char *mvfx_roundedcorners_cairo_version(void) { return cairo_mask_library_version(); }

int mvfx_roundedcorners_compose_a420(const mvfx_planar_frame *i420_in, const uint8_t *mask_device,
                                     uint32_t mask_stride, const mvfx_planar_frame *a420_out, mvfx_stream stream)
{
    if (!i420_in || !a420_out || !mask_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: NULL argument");
    if (i420_in->format != MVFX_FORMAT_I420 || a420_out->format != MVFX_FORMAT_A420)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "roundedcorners: I420 in, A420 out (border/imp.rs:345-365)");
    if (i420_in->width != a420_out->width || i420_in->height != a420_out->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "roundedcorners: input and output sizes differ");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t w = i420_in->width, h = i420_in->height, cw = (w + 1) / 2, ch = (h + 1) / 2;
    hipStream_t st = as_stream(stream);
    PlaneCopies pc{};
    for (int p = 0; p < 3; p++) {
        if (!i420_in->data[p] || !a420_out->data[p])
            return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: NULL plane %d", p);
        const uint32_t rb = p == 0 ? w : cw, rows = p == 0 ? h : ch;
        if (i420_in->stride[p] < rb || a420_out->stride[p] < rb)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: plane %d stride smaller than its row", p);
        pc.p[p] = PlaneCopy{static_cast<const uint8_t *>(i420_in->data[p]), static_cast<uint8_t *>(a420_out->data[p]), rb,
                            i420_in->stride[p], a420_out->stride[p], rows};
    }
    if (!a420_out->data[3] || a420_out->stride[3] < w || mask_stride < w)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: bad alpha plane");
    pc.p[3] = PlaneCopy{mask_device, static_cast<uint8_t *>(a420_out->data[3]), w, mask_stride, a420_out->stride[3], h};
    return launch_copy_planes(pc, 4, st);
}

} // extern "C"
