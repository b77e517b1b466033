// videocompare hash-algo = mean / gradient / vertgradient / doublegradient for gfx950
// (HashAlg::{Mean,Gradient,VertGradient,DoubleGradient}, video/videofx/src/videocompare/hashed_image.rs:89-107
//  -> image_hasher 3.1.1 `hash_image` on image 0.25.10: to_grayscale, imageops::resize(.., Lanczos3), then
//  compare-to-mean / neighbour compares on the 8x8 / 9x8 / 8x9 / 5x5 bytes).  Neither crate is under
//  /root/reference: PARITY UNPINNED against them; bit-exact against oracle/videofx_oracle.c.
//
// The resampler accumulates `t += px * w` in f32 in tap order, so each output value is one sequential chain of up
// to 6 * (size / out) additions whose roundings all matter (a different order flips u8 values on rounding
// boundaries).  What is parallel is everything around the chain:
//   vsample_kernel   one lane per (column, output row): the lane walks its column's taps in order; the wave reads
//                    64 consecutive pixels per tap (coalesced), integer Rec.709 luma on the fly, tap weights staged
//                    through LDS (every lane of the group needs the same weight at the same time: broadcast reads).
//                    8K frame -> 8 rows x 7680 columns = 61 440 chains of <= 3 240 taps.
//   hsample_kernel   one workgroup per output pixel: 256 lanes form the products row[i] * w[i] into LDS, lane 0
//                    adds them in order (the chain), clamps, rounds half away from zero, stores the byte.
//   gray_kernel      the copy path of imageops::resize (frame already has the target size).
// Tap tables come from the host (host/lanczos.cpp), cached per thread and frame size.
#include "mvfx_internal.h"

#include "lanczos.h"

#include <cstring>
#include <vector>

namespace mvfx {
namespace {

constexpr int kVBlock = 256;
constexpr uint32_t kVChunk = 2048;  // weights staged per LDS refill (8 KiB)
constexpr int kHBlock = 256;
constexpr uint32_t kHChunk = 4096;  // products per LDS refill (16 KiB)
constexpr uint32_t kMaxOut = 64;    // largest resize target per axis (the hashes need <= 9)

struct AxisDev {
    uint32_t left[kMaxOut], count[kMaxOut], offset[kMaxOut];
};

// image 0.25 color.rs rgb_to_luma for u8: (2126 r + 7152 g + 722 b) / 10000 in u32
__device__ __forceinline__ uint32_t luma709(uint32_t r, uint32_t g, uint32_t b) { return (2126u * r + 7152u * g + 722u * b) / 10000u; }

template <int BPP, bool DWORD>
__device__ __forceinline__ uint32_t load_luma(const uint8_t *p)
{
    if constexpr (BPP == 4 && DWORD) {
        const uint32_t px = *reinterpret_cast<const uint32_t *>(p);
        return luma709(px & 0xffu, (px >> 8) & 0xffu, (px >> 16) & 0xffu);
    } else {
        return luma709(p[0], p[1], p[2]);
    }
}


// vsample_block_kernel: the vertical pass for targets of <= 16 rows (every hash needs <= 9) reading the frame ONCE.
// A 1024-lane workgroup owns kVCols columns for ALL output rows.  Per chunk of kVRows input rows:
//   produce  all 16 waves load their pixels of the NEXT chunk (issued before the adds of the current one, so the HBM
//            round trip overlaps them), convert to integer Rec.709 luma and park it in LDS as f32;
//   add      wave `oy` (< nh) walks the rows of the chunk that fall into its tap window, lane = column:
//            t = t + G[row][column] * w[row - left], the crate's order; the weight index is wave-uniform (scalar loads).
// 8K frame, vertical pass: 220 us for the first version (vsample_kernel below: one lane per (column, output row), every
// row re-read by the ~6 output rows whose windows cover it, one wave per SIMD, one HBM round trip per 8 taps) and for
// this kernel while the adders fetched their weights from global memory; 128 us with the weights staged in LDS and 16
// taps read before they are added.  What is left is adder VALU time: 9 chains x 3 240 taps x 3 instructions per 32
// columns, on the 5 SIMD-slots the adder waves of a workgroup occupy.
constexpr int kVBlockAll = 1024;
constexpr uint32_t kVCols = 32;   // columns per workgroup (128-byte row segments for RGBA)
constexpr uint32_t kVRows = 128;  // input rows per chunk: G = 128 x 32 f32 = 16 KiB
constexpr uint32_t kVPasses = kVRows * kVCols / kVBlockAll; // loads per lane per chunk (4)

template <int BPP, bool DWORD>
__global__ __launch_bounds__(kVBlockAll) void vsample_block_kernel(const uint8_t *__restrict__ plane, uint64_t stride, uint32_t width,
                                                                   uint32_t height, uint32_t nh, AxisDev ax,
                                                                   const float *__restrict__ weights, float *__restrict__ tmp)
{
    __shared__ float G[kVRows][kVCols];                       // luma of the chunk, f32
    __shared__ __attribute__((aligned(16))) float W[kVBlockAll / 64][kVRows]; // weights of the chunk per output row, 0 outside its window
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    // producer role: lane -> (row within pass, column) for pixels, (output row, row of the chunk) for weights
    const uint32_t pc = threadIdx.x % kVCols, pr = threadIdx.x / kVCols; // pr in 0..31
    const uint32_t x = blockIdx.x * kVCols + pc;
    const bool px_ok = x < width;
    const uint8_t *col = plane + (uint64_t)(px_ok ? x : 0) * BPP;
    constexpr uint32_t kWPasses = (kVBlockAll / 64) * kVRows / kVBlockAll; // 2
    // adder role: wave a owns the output rows 2a (lanes 0..31) and 2a+1 (lanes 32..63), lane & 31 = column: neighbouring
    // rows share 5/6 of their tap windows, so the wave walks the union once with all 64 lanes busy
    const uint32_t a_oy = 2 * wave + (lane >> 5), a_col = lane & (kVCols - 1);
    const bool adder = a_oy < nh;
    const uint32_t oy_lo = min(2 * wave, nh - 1), oy_hi = min(2 * wave + 1, nh - 1);
    const uint32_t u_begin = ax.left[oy_lo], u_end = ax.left[oy_hi] + ax.count[oy_hi]; // union of the two windows (wave-uniform)
    const bool adder_wave = 2 * wave < nh;
    float t = 0.0f;

    uint32_t g[kVPasses];
    float wv[kWPasses];
    // the weight slots of this lane: (output row, row of the chunk) never change, so their window is looked up once
    uint32_t w_left[kWPasses], w_count[kWPasses], w_row[kWPasses];
    const float *w_ptr[kWPasses];
#pragma unroll
    for (uint32_t k = 0; k < kWPasses; k++) {
        const uint32_t idx = threadIdx.x + kVBlockAll * k, oy = idx / kVRows;
        w_row[k] = idx % kVRows;
        w_left[k] = oy < nh ? ax.left[oy] : 0;
        w_count[k] = oy < nh ? ax.count[oy] : 0;
        w_ptr[k] = weights + (oy < nh ? ax.offset[oy] : 0);
    }
    // All loads are unconditional (clamped addresses, results masked afterwards): a load inside a divergent branch gets
    // its own s_waitcnt, which serialised the six loads of a chunk into six memory round trips (220 -> 128 us was the
    // LDS staging, 128 -> see DESIGN.md was this).
    auto load_chunk = [&](uint32_t base) {
#pragma unroll
        for (uint32_t k = 0; k < kVPasses; k++) {
            const uint32_t r = base + pr + (kVBlockAll / kVCols) * k;
            const uint32_t v = load_luma<BPP, DWORD>(col + (uint64_t)min(r, height - 1) * stride);
            g[k] = (px_ok && r < height) ? v : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kWPasses; k++) {
            const uint32_t rel = base + w_row[k] - w_left[k]; // wraps for rows above the window: fails the unsigned test
            const float v = w_ptr[k][w_count[k] ? min(rel, w_count[k] - 1) : 0];
            // a weight of exactly 0 outside the tap window: t + G * 0 == t, so the adders run whole chunks
            wv[k] = rel < w_count[k] ? v : 0.0f;
        }
    };
    load_chunk(0);
    for (uint32_t base = 0; base < height; base += kVRows) {
#pragma unroll
        for (uint32_t k = 0; k < kVPasses; k++) G[pr + (kVBlockAll / kVCols) * k][pc] = (float)g[k];
#pragma unroll
        for (uint32_t k = 0; k < kWPasses; k++) (&W[0][0])[threadIdx.x + kVBlockAll * k] = wv[k];
        __syncthreads();
        if (base + kVRows < height) load_chunk(base + kVRows); // in flight while the adders run
        if (adder_wave && base < u_end && base + kVRows > u_begin) {
            const uint32_t w_row = adder ? a_oy : 2 * wave; // idle upper half (odd nh): reads a valid row, result dropped
            for (uint32_t i = 0; i < kVRows; i += 16) { // 16 LDS pixel reads + 4 weight reads issued, then the 16 chained additions
                float gv[16];
                float4 wq[4];
#pragma unroll
                for (int k = 0; k < 16; k++) gv[k] = G[i + k][a_col];
#pragma unroll
                for (int k = 0; k < 4; k++) wq[k] = *reinterpret_cast<const float4 *>(&W[w_row][i + 4 * k]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    t = t + gv[4 * k] * wq[k].x; t = t + gv[4 * k + 1] * wq[k].y;
                    t = t + gv[4 * k + 2] * wq[k].z; t = t + gv[4 * k + 3] * wq[k].w;
                }
            }
        }
        __syncthreads();
    }
    const uint32_t ax_x = blockIdx.x * kVCols + a_col;
    if (adder && ax_x < width) tmp[(uint64_t)a_oy * width + ax_x] = t;
}

template <int BPP, bool DWORD>
__global__ __launch_bounds__(kVBlock) void vsample_kernel(const uint8_t *plane, uint64_t stride, uint32_t width,
                                                          AxisDev ax, const float *weights, float *tmp)
{
    __shared__ float lw[kVChunk];
    const uint32_t oy = blockIdx.y;
    const uint32_t x = blockIdx.x * kVBlock + threadIdx.x;
    const uint32_t n = ax.count[oy];
    const float *w = weights + ax.offset[oy];
    const uint8_t *col = plane + (uint64_t)ax.left[oy] * stride + (uint64_t)(x < width ? x : 0) * BPP;
    float t = 0.0f;
    for (uint32_t base = 0; base < n; base += kVChunk) {
        const uint32_t m = min(kVChunk, n - base);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += kVBlock) lw[i] = w[base + i];
        __syncthreads();
        if (x < width) {
            const uint8_t *p = col + (uint64_t)base * stride;
            uint32_t i = 0;
            for (; i + 8 <= m; i += 8) { // 8 independent loads in flight, then the 8 chained additions
                uint32_t g[8];
#pragma unroll
                for (int k = 0; k < 8; k++) g[k] = load_luma<BPP, DWORD>(p + (uint64_t)(i + k) * stride);
#pragma unroll
                for (int k = 0; k < 8; k++) t = t + (float)g[k] * lw[i + k];
            }
            for (; i < m; i++) t = t + (float)load_luma<BPP, DWORD>(p + (uint64_t)i * stride) * lw[i];
        }
    }
    if (x < width) tmp[(uint64_t)oy * width + x] = t;
}

__global__ __launch_bounds__(kHBlock) void hsample_kernel(const float *tmp, uint32_t width, AxisDev ax,
                                                          const float *weights, uint8_t *out, uint32_t nw)
{
    __shared__ __attribute__((aligned(16))) float prod[kHChunk];
    const uint32_t ox = blockIdx.x, oy = blockIdx.y;
    const uint32_t n = ax.count[ox];
    const float *w = weights + ax.offset[ox];
    const float *row = tmp + (uint64_t)oy * width + ax.left[ox];
    float t = 0.0f;
    for (uint32_t base = 0; base < n; base += kHChunk) {
        const uint32_t m = min(kHChunk, n - base);
        __syncthreads();
        // products: four unconditional (clamped) load pairs per lane and pass, zero-padded to a multiple of 16 taps
        // (t + 0 == t), so the adder below runs whole batches
        const uint32_t mp = (m + 15) & ~15u;
        for (uint32_t i = threadIdx.x; i < mp; i += 4 * kHBlock) {
            float a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t idx = min(i + k * kHBlock, m - 1);
                a[k] = row[base + idx];
                b[k] = w[base + idx];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t idx = i + k * kHBlock;
                if (idx < mp) prod[idx] = idx < m ? a[k] * b[k] : 0.0f;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { // the chain: the next 16 products are read from LDS while the current 16 are added
            float4 q[4], nx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = *reinterpret_cast<const float4 *>(&prod[4 * k]);
            for (uint32_t i = 0; i < mp; i += 16) {
                const uint32_t j = i + 16 < mp ? i + 16 : i;
#pragma unroll
                for (int k = 0; k < 4; k++) nx[k] = *reinterpret_cast<const float4 *>(&prod[j + 4 * k]);
#pragma unroll
                for (int k = 0; k < 4; k++) { t = t + q[k].x; t = t + q[k].y; t = t + q[k].z; t = t + q[k].w; }
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = nx[k];
            }
        }
    }
    if (threadIdx.x == 0) {
        t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);  // image's clamp(): NaN passes through
        const float r = roundf(t);                          // FloatNearest: half away from zero
        out[oy * nw + ox] = r != r ? (uint8_t)0 : (uint8_t)r;
    }
}

template <int BPP>
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t *plane, uint64_t stride, uint32_t width, uint32_t height, uint8_t *out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t y = i / width, x = i % width;
    out[i] = (uint8_t)load_luma<BPP, false>(plane + (uint64_t)y * stride + (uint64_t)x * BPP);
}

// Per-thread plan cache: tap tables on the device for (width, height) -> (nw, nh), plus the f32 row buffer and the
// output bytes.  The four hash algorithms need four targets; a new frame size replaces the oldest plan.
struct Plan {
    uint32_t w = 0, h = 0, nw = 0, nh = 0;
    int device = -1;
    AxisDev v{}, hz{};
    float *weights = nullptr; // vertical weights followed by horizontal weights
    size_t v_floats = 0;
    float *tmp = nullptr;     // nh x w
    uint8_t *out = nullptr;   // nw x nh
    uint64_t stamp = 0;
    void release()
    {
        if (weights) (void)hipFree(weights);
        if (tmp) (void)hipFree(tmp);
        if (out) (void)hipFree(out);
        weights = nullptr; tmp = nullptr; out = nullptr; w = h = nw = nh = 0;
    }
};
struct PlanCache {
    Plan plans[4];
    uint64_t clock = 0;
    ~PlanCache() { for (Plan &p : plans) p.release(); }
};
thread_local PlanCache t_plans;

int get_plan(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, hipStream_t st, Plan **out)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    Plan *victim = &t_plans.plans[0];
    for (Plan &p : t_plans.plans) {
        if (p.w == w && p.h == h && p.nw == nw && p.nh == nh && p.device == dev) {
            p.stamp = ++t_plans.clock;
            *out = &p;
            return MVFX_OK;
        }
        if (p.stamp < victim->stamp) victim = &p;
    }
    Plan &p = *victim;
    MVFX_HIP_TRY(hipStreamSynchronize(st)); // an evicted plan may still be read by queued kernels
    p.release();
    const LanczosAxis va = lanczos3_axis(h, nh), ha = lanczos3_axis(w, nw);
    for (uint32_t o = 0; o < nh; o++) { p.v.left[o] = va.left[o]; p.v.count[o] = va.count[o]; p.v.offset[o] = va.offset[o]; }
    for (uint32_t o = 0; o < nw; o++) { p.hz.left[o] = ha.left[o]; p.hz.count[o] = ha.count[o]; p.hz.offset[o] = ha.offset[o]; }
    p.v_floats = va.weights.size();
    const size_t total = va.weights.size() + ha.weights.size();
    if (hipMalloc(&p.weights, total * sizeof(float)) != hipSuccess || hipMalloc(&p.tmp, (size_t)nh * w * sizeof(float)) != hipSuccess ||
        hipMalloc(&p.out, (size_t)nw * nh) != hipSuccess) {
        p.release();
        return fail(MVFX_ERR_OUT_OF_MEMORY, "videocompare: device allocation for the %ux%u -> %ux%u resize plan failed", w, h, nw, nh);
    }
    MVFX_HIP_TRY(hipMemcpy(p.weights, va.weights.data(), va.weights.size() * sizeof(float), hipMemcpyHostToDevice));
    MVFX_HIP_TRY(hipMemcpy(p.weights + p.v_floats, ha.weights.data(), ha.weights.size() * sizeof(float), hipMemcpyHostToDevice));
    p.w = w; p.h = h; p.nw = nw; p.nh = nh; p.device = dev;
    p.stamp = ++t_plans.clock;
    *out = &p;
    return MVFX_OK;
}

// to_grayscale + imageops::resize(gray, nw, nh, Lanczos3) of a device frame -> nw*nh bytes on the host (synchronous)
int gray_resize_impl(const mvfx_frame *frame, uint32_t nw, uint32_t nh, uint8_t *out_host, hipStream_t st)
{
    if (!frame || !out_host)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: NULL argument");
    if (frame->format != MVFX_FORMAT_RGB && frame->format != MVFX_FORMAT_RGBA)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "videocompare: format %d is not RGB / RGBA (videocompare/imp.rs:160-162)", frame->format);
    if (int rc = check_packed_frame(frame, "videocompare"); rc != MVFX_OK) return rc;
    if (nw == 0 || nh == 0 || nw > kMaxOut || nh > kMaxOut)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: resize target %ux%u outside 1..%u", nw, nh, kMaxOut);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const uint32_t w = frame->width, h = frame->height;
    if (w == 0 || h == 0) { // imageops::resize of an empty image: a zeroed buffer
        std::memset(out_host, 0, (size_t)nw * nh);
        return MVFX_OK;
    }
    const uint8_t *plane = static_cast<const uint8_t *>(frame->data);
    const bool rgba = frame->format == MVFX_FORMAT_RGBA;
    if (nw == w && nh == h) { // same dimensions: imageops::resize copies, no resampling
        void *scratch = nullptr;
        if (int rc = host_scratch((size_t)w * h, 2, &scratch); rc != MVFX_OK) return rc;
        uint8_t *out = static_cast<uint8_t *>(scratch);
        const dim3 grid((w * h + 255) / 256);
        if (rgba) MVFX_LAUNCH(gray_kernel<4>, grid, dim3(256), 0, st, plane, (uint64_t)frame->stride, w, h, out);
        else MVFX_LAUNCH(gray_kernel<3>, grid, dim3(256), 0, st, plane, (uint64_t)frame->stride, w, h, out);
        MVFX_HIP_TRY(hipGetLastError());
        MVFX_HIP_TRY(hipMemcpyAsync(out_host, out, (size_t)w * h, hipMemcpyDeviceToHost, st));
        MVFX_HIP_TRY(hipStreamSynchronize(st));
        return MVFX_OK;
    }
    Plan *plan = nullptr;
    if (int rc = get_plan(w, h, nw, nh, st, &plan); rc != MVFX_OK) return rc;
    const bool dword = rgba && ((reinterpret_cast<uintptr_t>(plane) | frame->stride) & 3) == 0;
    static_assert(kVCols == 32, "the adder waves pair two output rows of 32 columns");
    if (nh <= kVBlockAll / 64) { // one pass over the frame, all output rows at once
        const dim3 bgrid((w + kVCols - 1) / kVCols);
        if (rgba && dword)
            MVFX_LAUNCH((vsample_block_kernel<4, true>), bgrid, dim3(kVBlockAll), 0, st, plane, (uint64_t)frame->stride, w, h, nh, plan->v, plan->weights, plan->tmp);
        else if (rgba)
            MVFX_LAUNCH((vsample_block_kernel<4, false>), bgrid, dim3(kVBlockAll), 0, st, plane, (uint64_t)frame->stride, w, h, nh, plan->v, plan->weights, plan->tmp);
        else
            MVFX_LAUNCH((vsample_block_kernel<3, false>), bgrid, dim3(kVBlockAll), 0, st, plane, (uint64_t)frame->stride, w, h, nh, plan->v, plan->weights, plan->tmp);
    } else {
        const dim3 vgrid((w + kVBlock - 1) / kVBlock, nh);
        if (rgba && dword)
            MVFX_LAUNCH((vsample_kernel<4, true>), vgrid, dim3(kVBlock), 0, st, plane, (uint64_t)frame->stride, w, plan->v, plan->weights, plan->tmp);
        else if (rgba)
            MVFX_LAUNCH((vsample_kernel<4, false>), vgrid, dim3(kVBlock), 0, st, plane, (uint64_t)frame->stride, w, plan->v, plan->weights, plan->tmp);
        else
            MVFX_LAUNCH((vsample_kernel<3, false>), vgrid, dim3(kVBlock), 0, st, plane, (uint64_t)frame->stride, w, plan->v, plan->weights, plan->tmp);
    }
    MVFX_LAUNCH(hsample_kernel, dim3(nw, nh), dim3(kHBlock), 0, st, plan->tmp, w, plan->hz, plan->weights + plan->v_floats, plan->out, nw);
    MVFX_HIP_TRY(hipGetLastError());
    MVFX_HIP_TRY(hipMemcpyAsync(out_host, plan->out, (size_t)nw * nh, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return MVFX_OK;
}

// image_hasher HashAlg::resize_dimensions for the default 8x8 hash size
bool resize_dimensions(int algo, uint32_t *rw, uint32_t *rh)
{
    switch (algo) {
    case MVFX_HASH_MEAN: *rw = 8; *rh = 8; return true;
    case MVFX_HASH_GRADIENT: *rw = 9; *rh = 8; return true;
    case MVFX_HASH_VERTGRADIENT: *rw = 8; *rh = 9; return true;
    case MVFX_HASH_DOUBLEGRADIENT: *rw = 5; *rh = 5; return true;
    default: return false;
    }
}

// mean_hash_u8 / gradient_hash / vert_gradient_hash / double_gradient_hash on the resized bytes; bit k = the k-th
// bool of the crate's iterator (the Hamming distance does not depend on the packing)
uint64_t hash_bits(int algo, const uint8_t *px, uint32_t rw, uint32_t rh, uint32_t *n_bits)
{
    uint64_t h = 0;
    uint32_t k = 0;
    if (algo == MVFX_HASH_MEAN) {
        uint32_t sum = 0;
        for (uint32_t i = 0; i < rw * rh; i++) sum += px[i];
        const uint8_t mean = (uint8_t)(sum / (rw * rh));
        for (uint32_t i = 0; i < rw * rh; i++, k++)
            if (px[i] >= mean) h |= 1ull << k;
    }
    if (algo == MVFX_HASH_GRADIENT || algo == MVFX_HASH_DOUBLEGRADIENT)
        for (uint32_t y = 0; y < rh; y++)
            for (uint32_t x = 0; x + 1 < rw; x++, k++)
                if (px[y * rw + x] < px[y * rw + x + 1]) h |= 1ull << k;
    if (algo == MVFX_HASH_VERTGRADIENT || algo == MVFX_HASH_DOUBLEGRADIENT)
        for (uint32_t x = 0; x < rw; x++)
            for (uint32_t y = 0; y + 1 < rh; y++, k++)
                if (px[y * rw + x] < px[(y + 1) * rw + x]) h |= 1ull << k;
    *n_bits = k;
    return h;
}

int image_hash_impl(const mvfx_frame *frame, int algo, uint64_t *hash_out, uint32_t *n_bits_out, hipStream_t st)
{
    if (!hash_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: NULL output");
    uint32_t rw = 0, rh = 0;
    if (!resize_dimensions(algo, &rw, &rh))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: hash algorithm %d is not mean / gradient / vertgradient / doublegradient", algo);
    uint8_t px[kMaxOut * kMaxOut];
    if (int rc = gray_resize_impl(frame, rw, rh, px, st); rc != MVFX_OK) return rc;
    uint32_t n = 0;
    *hash_out = hash_bits(algo, px, rw, rh, &n);
    if (n_bits_out) *n_bits_out = n;
    return MVFX_OK;
}

int upload_frame(const mvfx_frame *frame, int slot, mvfx_frame *dev_frame, hipStream_t st)
{
    if (!frame)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: NULL frame");
    if (int rc = check_packed_frame(frame, "videocompare"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    const size_t bytes = (size_t)frame->stride * frame->height;
    void *dev = nullptr;
    if (int rc = host_scratch(bytes ? bytes : 16, slot, &dev); rc != MVFX_OK) return rc;
    if (bytes)
        MVFX_HIP_TRY(hipMemcpyAsync(dev, frame->data, bytes, hipMemcpyHostToDevice, st));
    *dev_frame = *frame;
    dev_frame->data = dev;
    return MVFX_OK;
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_image_gray_resize_lanczos3(const mvfx_frame *frame, uint32_t new_width, uint32_t new_height, uint8_t *out_host,
                                    mvfx_stream stream)
{
    return gray_resize_impl(frame, new_width, new_height, out_host, as_stream(stream));
}

int mvfx_image_hash(const mvfx_frame *frame, int32_t hash_algo, uint64_t *hash_out, uint32_t *n_bits_out, mvfx_stream stream)
{
    return image_hash_impl(frame, hash_algo, hash_out, n_bits_out, as_stream(stream));
}

int mvfx_image_hash_host(const mvfx_frame *frame, int32_t hash_algo, uint64_t *hash_out, uint32_t *n_bits_out)
{
    mvfx_frame d;
    hipStream_t st = host_stream();
    if (int rc = upload_frame(frame, 0, &d, st); rc != MVFX_OK) return rc;
    return image_hash_impl(&d, hash_algo, hash_out, n_bits_out, st);
}

// HasherEngine::from(algo) + hash_image x2 + compare (hashed_image.rs:24-107) for any hash-algo value
int mvfx_videocompare_distance_algo(const mvfx_frame *reference_frame, const mvfx_frame *other_frame, int32_t hash_algo,
                                    double *distance_out, mvfx_stream stream)
{
    if (!reference_frame || !other_frame || !distance_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: NULL argument");
    if (hash_algo == MVFX_HASH_BLOCKHASH)
        return mvfx_videocompare_distance(reference_frame, other_frame, distance_out, stream);
    if (hash_algo == MVFX_HASH_DSSIM)
        return mvfx_ssim_distance(reference_frame, other_frame, distance_out, stream);
    if (reference_frame->width != other_frame->width || reference_frame->height != other_frame->height)
        return fail(MVFX_ERR_NOT_NEGOTIATED, "Video streams do not have the same sizes (videocompare/imp.rs:337-346)");
    uint64_t a = 0, b = 0;
    if (int rc = image_hash_impl(reference_frame, hash_algo, &a, nullptr, as_stream(stream)); rc != MVFX_OK) return rc;
    if (int rc = image_hash_impl(other_frame, hash_algo, &b, nullptr, as_stream(stream)); rc != MVFX_OK) return rc;
    *distance_out = (double)mvfx_hash_distance(a, b); // hashed_image.rs:70
    return MVFX_OK;
}

} // extern "C"
