// tools/probe_copy.hip -- HBM ceilings on this box for the access shapes the path uses:
// (a) out-of-place uint4 copy, (b) in-place read-modify-write (what hsvfilter does), (c) read-only.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_copy.hip -o /tmp/probe_copy && /tmp/probe_copy
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_rmw(uint4 *buf, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = buf[i];
        v.x ^= 0x01010101u; v.y += 3u; v.z ^= v.x; v.w += v.y;
        buf[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_read(const uint4 *in, unsigned *sink, size_t n)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)16 * 3840 * 2160 * 4 * 2; // 1.06 GB: 32 4K frames, far beyond the 256 MiB L3
    const size_t n = bytes / 16;
    uint4 *a, *b; unsigned *sink;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {2048, 8192, 65536}) {
        for (int which = 0; which < 3; which++) {
            float best = 1e30f, sum = 0;
            const int reps = 12;
            for (int r = 0; r < reps + 2; r++) {
                (void)hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n / 2);
                else if (which == 1) hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, a, n / 2);
                else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            const double moved = which == 2 ? (double)bytes : (double)bytes; // copy/rmw: n/2 elements read + written
            printf("grid %6d %-22s avg %.3f ms  %.0f GB/s (best %.0f GB/s)\n", grid,
                   which == 0 ? "copy 531MB->531MB" : which == 1 ? "in-place rmw 531MB" : "read-only 1.06GB",
                   sum / reps, moved / (sum / reps * 1e-3) / 1e9, moved / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
