// tools/probe_blockhash.hip -- access-pattern A/B for the videocompare block-sum reduction on 8K RGBA frames
// (7680x4320, 132.7 MB each; 4 frames rotate so reads come from HBM).  Every variant computes the same 64 sums.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_blockhash.hip -o tools/probe_blockhash.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bright(uint32_t px)
{
    return px < 0x01000000u ? 765u : __builtin_amdgcn_sad_u8(px & 0x00ffffffu, 0u, 0u);
}
__device__ __forceinline__ uint32_t bright4(u32x4 v) { return bright(v.x) + bright(v.y) + bright(v.z) + bright(v.w); }

__device__ __forceinline__ void wg_flush(uint32_t acc, uint32_t *dst)
{
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = ws[0] + ws[1] + ws[2] + ws[3]; if (t) atomicAdd(dst, t); }
}

// V0: round-1 kernel (grid chunks x 64, flat index with a 64-bit division per load)
__global__ __launch_bounds__(256) void v0(const uint8_t *plane, uint32_t bw, uint32_t bh, uint64_t stride, uint32_t chunks, uint32_t *sums)
{
    const uint32_t block = blockIdx.y, bx = block & 7, by = block >> 3;
    const uint32_t y0 = by * bh;
    uint32_t acc = 0;
    const uint32_t gpr = bw >> 2;
    const uint64_t total = (uint64_t)bh * gpr;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)chunks * 256) {
        const uint32_t r = (uint32_t)(i / gpr), gx = (uint32_t)(i % gpr);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(plane + (uint64_t)(y0 + r) * stride + ((uint64_t)bx * bw + gx * 4) * 4);
        acc += bright4(v);
    }
    wg_flush(acc, &sums[block]);
}

// V1: per-block workgroups, LX x RY lanes, K rows in flight per lane (rows `step` apart)
template <int K, bool NT>
__global__ __launch_bounds__(256) void v1(const uint8_t *plane, uint32_t bw, uint32_t bh, uint64_t stride, uint32_t lx_log2, uint32_t *sums)
{
    const uint32_t block = blockIdx.y, bx = block & 7, by = block >> 3;
    const uint32_t y0 = by * bh, y1 = y0 + bh;
    const uint32_t lx = 1u << lx_log2, ry = 256 >> lx_log2;
    const uint32_t tx = threadIdx.x & (lx - 1), ty = threadIdx.x >> lx_log2;
    const uint32_t step = gridDim.x * ry;
    const uint8_t *base = plane + (uint64_t)bx * bw * 4;
    const uint32_t units = bw >> 2;
    uint32_t acc = 0;
    uint32_t r = y0 + blockIdx.x * ry + ty;
    for (; r < y1 && y1 - r > (K - 1) * step; r += K * step)
        for (uint32_t u = tx; u < units; u += lx) {
            u32x4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (uint64_t)(r + k * step) * stride) + u;
                v[k] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int k = 0; k < K; k++) acc += bright4(v[k]);
        }
    for (; r < y1; r += step)
        for (uint32_t u = tx; u < units; u += lx) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (uint64_t)r * stride) + u;
            acc += bright4(NT ? __builtin_nontemporal_load(p) : *p);
        }
    wg_flush(acc, &sums[block]);
}

// V2: full-width rows.  Workgroup k owns rows [k*R, k*R+R) (contiguous R*stride bytes); lane t owns the 16-byte
// columns t, t+256, ... (J of them, J <= 8), each of which lies in ONE block column for every row => J register
// accumulators, flushed into 8 LDS sums (then 8 global atomics) when the block row changes or at the end.
template <int J, bool NT>
__global__ __launch_bounds__(256) void v2(const uint8_t *plane, uint32_t w, uint32_t h, uint64_t stride, uint32_t R, uint32_t *sums)
{
    const uint32_t units = w >> 2, gpr = w >> 5, bh = h >> 3;
    __shared__ uint32_t s8[8];
    if (threadIdx.x < 8) s8[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc[J];
    uint32_t bxj[J];
#pragma unroll
    for (int j = 0; j < J; j++) { acc[j] = 0; bxj[j] = (threadIdx.x + 256u * j) / gpr; }
    const uint32_t r0 = blockIdx.x * R, r1 = min(r0 + R, h);
    uint32_t by = r0 / bh;
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t nby = r / bh;
        if (nby != by) { // block row changes inside this workgroup's rows: flush
#pragma unroll
            for (int j = 0; j < J; j++) { if (threadIdx.x + 256u * j < units) atomicAdd(&s8[bxj[j]], acc[j]); acc[j] = 0; }
            __syncthreads();
            if (threadIdx.x < 8) { if (s8[threadIdx.x]) atomicAdd(&sums[by * 8 + threadIdx.x], s8[threadIdx.x]); s8[threadIdx.x] = 0; }
            __syncthreads();
            by = nby;
        }
        const u32x4 *row = reinterpret_cast<const u32x4 *>(plane + (uint64_t)r * stride);
        u32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t u = threadIdx.x + 256u * j;
            if (u < units) v[j] = NT ? __builtin_nontemporal_load(row + u) : row[u];
            else v[j] = u32x4{0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
        }
#pragma unroll
        for (int j = 0; j < J; j++) acc[j] += bright4(v[j]);
    }
    // wave-level combine of equal block columns, then LDS, then global
#pragma unroll
    for (int j = 0; j < J; j++) {
        const bool live = threadIdx.x + 256u * j < units;
        const uint32_t b = live ? bxj[j] : 0xffffffffu;
        const uint32_t first = __builtin_amdgcn_readfirstlane(b);
        if (__all(b == first)) {
            uint32_t a = acc[j];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if ((threadIdx.x & 63) == 0 && first != 0xffffffffu) atomicAdd(&s8[first], a);
        } else if (live) {
            atomicAdd(&s8[b], acc[j]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 && s8[threadIdx.x]) atomicAdd(&sums[by * 8 + threadIdx.x], s8[threadIdx.x]);
}

// V3: like V2 but a persistent grid: workgroup k takes rows k, k+G, k+2G ... (the whole grid reads a contiguous
// window that moves through the frame); per-lane accumulators flushed when the block row changes.
template <int J, bool NT>
__global__ __launch_bounds__(256) void v3(const uint8_t *plane, uint32_t w, uint32_t h, uint64_t stride, uint32_t *sums)
{
    const uint32_t units = w >> 2, gpr = w >> 5, bh = h >> 3;
    __shared__ uint32_t s8[8];
    if (threadIdx.x < 8) s8[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc[J];
    uint32_t bxj[J];
#pragma unroll
    for (int j = 0; j < J; j++) { acc[j] = 0; bxj[j] = (threadIdx.x + 256u * j) / gpr; }
    uint32_t by = blockIdx.x / bh;
    for (uint32_t r = blockIdx.x; r < h; r += gridDim.x) {
        const uint32_t nby = r / bh;
        if (nby != by) {
#pragma unroll
            for (int j = 0; j < J; j++) {
                uint32_t a = acc[j];
                const bool live = threadIdx.x + 256u * j < units;
                const uint32_t b = live ? bxj[j] : 0xffffffffu;
                const uint32_t first = __builtin_amdgcn_readfirstlane(b);
                if (__all(b == first)) {
                    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
                    if ((threadIdx.x & 63) == 0 && first != 0xffffffffu) atomicAdd(&s8[first], a);
                } else if (live) atomicAdd(&s8[b], a);
                acc[j] = 0;
            }
            __syncthreads();
            if (threadIdx.x < 8) { if (s8[threadIdx.x]) atomicAdd(&sums[by * 8 + threadIdx.x], s8[threadIdx.x]); s8[threadIdx.x] = 0; }
            __syncthreads();
            by = nby;
        }
        const u32x4 *row = reinterpret_cast<const u32x4 *>(plane + (uint64_t)r * stride);
        u32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t u = threadIdx.x + 256u * j;
            if (u < units) v[j] = NT ? __builtin_nontemporal_load(row + u) : row[u];
            else v[j] = u32x4{0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
        }
#pragma unroll
        for (int j = 0; j < J; j++) acc[j] += bright4(v[j]);
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
        uint32_t a = acc[j];
        const bool live = threadIdx.x + 256u * j < units;
        const uint32_t b = live ? bxj[j] : 0xffffffffu;
        const uint32_t first = __builtin_amdgcn_readfirstlane(b);
        if (__all(b == first)) {
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if ((threadIdx.x & 63) == 0 && first != 0xffffffffu) atomicAdd(&s8[first], a);
        } else if (live) atomicAdd(&s8[b], a);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s8[threadIdx.x]) atomicAdd(&sums[by * 8 + threadIdx.x], s8[threadIdx.x]);
}

// V4: plain flat read of the same bytes (no block bookkeeping): the ceiling of this access shape
__global__ __launch_bounds__(256) void v4(const u32x4 *in, size_t n, uint32_t *sums)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += bright4(in[i]);
    wg_flush(acc, &sums[blockIdx.x & 63]);
}


// ---- no same-address atomics: every workgroup stores its partial, a second small launch adds them up
template <int K, bool NT>
__global__ __launch_bounds__(256) void v5(const uint8_t *plane, uint32_t bw, uint32_t bh, uint64_t stride, uint32_t lx_log2, uint32_t *partials)
{
    const uint32_t block = blockIdx.y, bx = block & 7, by = block >> 3;
    const uint32_t y0 = by * bh, y1 = y0 + bh;
    const uint32_t lx = 1u << lx_log2, ry = 256 >> lx_log2;
    const uint32_t tx = threadIdx.x & (lx - 1), ty = threadIdx.x >> lx_log2;
    const uint32_t step = gridDim.x * ry;
    const uint8_t *base = plane + (uint64_t)bx * bw * 4;
    const uint32_t units = bw >> 2;
    uint32_t acc = 0;
    uint32_t r = y0 + blockIdx.x * ry + ty;
    for (; r < y1 && y1 - r > (K - 1) * step; r += K * step)
        for (uint32_t u = tx; u < units; u += lx) {
            u32x4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (uint64_t)(r + k * step) * stride) + u;
                v[k] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int k = 0; k < K; k++) acc += bright4(v[k]);
        }
    for (; r < y1; r += step)
        for (uint32_t u = tx; u < units; u += lx) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (uint64_t)r * stride) + u;
            acc += bright4(NT ? __builtin_nontemporal_load(p) : *p);
        }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[block * gridDim.x + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(64) void reduce_partials(const uint32_t *partials, uint32_t n_per_block, uint32_t *sums)
{
    // one wave per block: lanes stride over the partials, shuffle reduce
    const uint32_t block = blockIdx.x;
    uint32_t a = 0;
    for (uint32_t i = threadIdx.x; i < n_per_block; i += 64) a += partials[block * n_per_block + i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (threadIdx.x == 0) sums[block] = a;
}

// V6: V2's row-contiguous workgroups inside one block row (grid x = row groups of the block row, y = block row);
// partial = 8 values per workgroup, layout partials[by*8+bx][x]
template <int J, bool NT>
__global__ __launch_bounds__(256) void v6(const uint8_t *plane, uint32_t w, uint32_t h, uint64_t stride, uint32_t R, uint32_t *partials)
{
    const uint32_t units = w >> 2, gpr = w >> 5, bh = h >> 3;
    __shared__ uint32_t s8[8];
    if (threadIdx.x < 8) s8[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc[J];
#pragma unroll
    for (int j = 0; j < J; j++) acc[j] = 0;
    const uint32_t by = blockIdx.y;
    const uint32_t r0 = by * bh + blockIdx.x * R, r1 = min(r0 + R, (by + 1) * bh);
    for (uint32_t r = r0; r < r1; r++) {
        const u32x4 *row = reinterpret_cast<const u32x4 *>(plane + (uint64_t)r * stride);
        u32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t u = threadIdx.x + 256u * j;
            if (u < units) v[j] = NT ? __builtin_nontemporal_load(row + u) : row[u];
            else v[j] = u32x4{0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
        }
#pragma unroll
        for (int j = 0; j < J; j++) acc[j] += bright4(v[j]);
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
        uint32_t a = acc[j];
        const bool live = threadIdx.x + 256u * j < units;
        const uint32_t b = live ? (threadIdx.x + 256u * j) / gpr : 0xffffffffu;
        const uint32_t first = __builtin_amdgcn_readfirstlane(b);
        if (__all(b == first)) {
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if ((threadIdx.x & 63) == 0 && first != 0xffffffffu) atomicAdd(&s8[first], a);
        } else if (live) atomicAdd(&s8[b], a);
    }
    __syncthreads();
    if (threadIdx.x < 8) partials[(by * 8 + threadIdx.x) * gridDim.x + blockIdx.x] = s8[threadIdx.x];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    const uint32_t W = 7680, H = 4320;
    const uint64_t stride = (uint64_t)W * 4;
    const size_t bytes = (size_t)stride * H;
    const int NF = 4;
    uint8_t *frames[NF];
    std::vector<uint8_t> host(bytes);
    uint64_t s = 0x5EED0001;
    for (size_t i = 0; i < bytes; i += 8) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; memcpy(&host[i], &z, 8); }
    for (int f = 0; f < NF; f++) { CK(hipMalloc(&frames[f], bytes)); host[f] ^= 0x55; CK(hipMemcpy(frames[f], host.data(), bytes, hipMemcpyHostToDevice)); }
    uint32_t *sums; CK(hipMalloc(&sums, 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t bw = W / 8, bh = H / 8;
    uint32_t ref[64]; bool have_ref = false;

    auto run = [&](const char *name, auto launch) -> int {
        // steady clocks first
        auto t0 = std::chrono::steady_clock::now();
        int k = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.4) { for (int i = 0; i < 20; i++) launch(frames[k++ % NF]); CK(hipDeviceSynchronize()); }
        const int iters = 200;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) launch(frames[i % NF]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemset(sums, 0, 256)); launch(frames[0]); CK(hipDeviceSynchronize());
        uint32_t got[64]; CK(hipMemcpy(got, sums, 256, hipMemcpyDeviceToHost));
        const char *ok = "";
        if (strncmp(name, "V4", 2) != 0) {
            if (!have_ref) { memcpy(ref, got, 256); have_ref = true; ok = "(reference)"; }
            else ok = memcmp(ref, got, 256) == 0 ? "sums ok" : "SUMS DIFFER";
        }
        printf("%-44s %7.2f us/frame  %6.0f GB/s  %s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9, ok);
        return 0;
    };
#define RUN(name, ...) if (run(name, [&](const uint8_t *p) { __VA_ARGS__; })) return 1

    RUN("V0 round-1 kernel (64 chunks x 64, div)", hipLaunchKernelGGL(v0, dim3(64, 64), dim3(256), 0, 0, p, bw, bh, stride, 64u, sums));
    RUN("V0 + memset", (void)hipMemsetAsync(sums, 0, 256, 0); hipLaunchKernelGGL(v0, dim3(64, 64), dim3(256), 0, 0, p, bw, bh, stride, 64u, sums));
    RUN("V1 K=4 NT   (135 x 64)", hipLaunchKernelGGL((v1<4, true>), dim3(135, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    RUN("V1 K=4 plain (135 x 64)", hipLaunchKernelGGL((v1<4, false>), dim3(135, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    RUN("V1 K=4 plain (34 x 64, 4 batches)", hipLaunchKernelGGL((v1<4, false>), dim3(34, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    RUN("V1 K=8 plain (68 x 64)", hipLaunchKernelGGL((v1<8, false>), dim3(68, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    RUN("V1 K=1 plain (34 x 64)", hipLaunchKernelGGL((v1<1, false>), dim3(34, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    RUN("V1 K=2 plain (34 x 64)", hipLaunchKernelGGL((v1<2, false>), dim3(34, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, sums));
    for (uint32_t R : {1u, 2u, 4u, 8u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "V2 rows-contiguous R=%u plain", R);
        RUN(nm, hipLaunchKernelGGL((v2<8, false>), dim3((H + R - 1) / R), dim3(256), 0, 0, p, W, H, stride, R, sums));
        snprintf(nm, sizeof nm, "V2 rows-contiguous R=%u NT", R);
        RUN(nm, hipLaunchKernelGGL((v2<8, true>), dim3((H + R - 1) / R), dim3(256), 0, 0, p, W, H, stride, R, sums));
    }
    for (uint32_t G : {512u, 1024u, 2048u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "V3 persistent rows G=%u plain", G);
        RUN(nm, hipLaunchKernelGGL((v3<8, false>), dim3(G), dim3(256), 0, 0, p, W, H, stride, sums));
        snprintf(nm, sizeof nm, "V3 persistent rows G=%u NT", G);
        RUN(nm, hipLaunchKernelGGL((v3<8, true>), dim3(G), dim3(256), 0, 0, p, W, H, stride, sums));
    }
    uint32_t *partials; CK(hipMalloc(&partials, 64 * 1024 * 4));
    for (uint32_t C : {34u, 68u, 135u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "V5 per-block K=4 NT partials (%u x 64) + reduce", C);
        RUN(nm, hipLaunchKernelGGL((v5<4, true>), dim3(C, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, partials);
                hipLaunchKernelGGL(reduce_partials, dim3(64), dim3(64), 0, 0, partials, C, sums));
    }
    RUN("V5 per-block K=8 NT partials (68 x 64) + reduce", hipLaunchKernelGGL((v5<8, true>), dim3(68, 64), dim3(256), 0, 0, p, bw, bh, stride, 8u, partials);
                hipLaunchKernelGGL(reduce_partials, dim3(64), dim3(64), 0, 0, partials, 68u, sums));
    for (uint32_t R : {1u, 2u, 3u, 4u, 6u, 8u}) {
        char nm[64];
        const uint32_t gx = (bh + R - 1) / R;
        snprintf(nm, sizeof nm, "V6 rows-contiguous R=%u NT partials + reduce", R);
        RUN(nm, hipLaunchKernelGGL((v6<8, true>), dim3(gx, 8), dim3(256), 0, 0, p, W, H, stride, R, partials);
                hipLaunchKernelGGL(reduce_partials, dim3(64), dim3(64), 0, 0, partials, gx, sums));
        snprintf(nm, sizeof nm, "V6 rows-contiguous R=%u plain partials + reduce", R);
        RUN(nm, hipLaunchKernelGGL((v6<8, false>), dim3(gx, 8), dim3(256), 0, 0, p, W, H, stride, R, partials);
                hipLaunchKernelGGL(reduce_partials, dim3(64), dim3(64), 0, 0, partials, gx, sums));
    }
    for (uint32_t G : {2048u, 8192u, 32400u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "V4 flat read ceiling G=%u", G);
        RUN(nm, hipLaunchKernelGGL(v4, dim3(G), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(p), bytes / 16, sums));
    }
    return 0;
}
