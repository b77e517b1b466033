// tools/probe_pmc_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, one kernel
// per access pattern this library uses.  MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports 1/2 of a wide coalesced 16 B/lane
// read stream; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern".  Every kernel touches a 1 GiB region (4x the 256 MiB Infinity Cache) exactly once.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_pmc_calib.hip -o tools/probe_pmc_calib.bin
//   rocprofv3 --pmc FETCH_SIZE -- tools/probe_pmc_calib.bin ; rocprofv3 --pmc WRITE_SIZE -- tools/probe_pmc_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void calib_read16(const u32x4 *p, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = p[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = 1; }
}
__global__ __launch_bounds__(256) void calib_read16_nt(const u32x4 *p, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = __builtin_nontemporal_load(p + i); if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = 1; }
}
__global__ __launch_bounds__(256) void calib_read8(const u32x2 *p, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x2 v = p[i]; if ((v.x ^ v.y) == 0x12345u) sink[0] = 1; }
}
__global__ __launch_bounds__(256) void calib_read4(const unsigned *p, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { if (p[i] == 0x12345u) sink[0] = 1; }
}
__global__ __launch_bounds__(256) void calib_read1(const unsigned char *p, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { if (p[i] == 0x77u) sink[0] = 1; }
}
// colordetect's pattern: one dword every 40 bytes (quality = 10 on RGBA): every 64-byte line of the region is touched
__global__ __launch_bounds__(256) void calib_read4_stride40(const unsigned char *p, size_t n_samples, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_samples) { if (*(const unsigned *)(p + i * 40) == 0x12345u) sink[0] = 1; }
}
__global__ __launch_bounds__(256) void calib_write16(u32x4 *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = {(unsigned)i, 1u, 2u, 3u}; p[i] = v; }
}
__global__ __launch_bounds__(256) void calib_write16_nt(u32x4 *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = {(unsigned)i, 1u, 2u, 3u}; __builtin_nontemporal_store(v, p + i); }
}
__global__ __launch_bounds__(256) void calib_write4(unsigned *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}
__global__ __launch_bounds__(256) void calib_write1(unsigned char *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (unsigned char)i;
}
__global__ __launch_bounds__(256) void calib_rmw16_nt(u32x4 *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = __builtin_nontemporal_load(p + i); v.x ^= 1u; __builtin_nontemporal_store(v, p + i); }
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned char *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 7, bytes);
    (void)hipDeviceSynchronize();
    auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    printf("# every kernel touches %zu bytes once (1 GiB region, 4x the Infinity Cache); 3 launches each\n", bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read16, grid(bytes / 16), dim3(256), 0, 0, (const u32x4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_read16_nt, grid(bytes / 16), dim3(256), 0, 0, (const u32x4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_read8, grid(bytes / 8), dim3(256), 0, 0, (const u32x2 *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(calib_read4, grid(bytes / 4), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(calib_read1, grid(bytes), dim3(256), 0, 0, (const unsigned char *)buf, bytes, sink);
        hipLaunchKernelGGL(calib_read4_stride40, grid(bytes / 40), dim3(256), 0, 0, (const unsigned char *)buf, bytes / 40, sink);
        hipLaunchKernelGGL(calib_write16, grid(bytes / 16), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16);
        hipLaunchKernelGGL(calib_write16_nt, grid(bytes / 16), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16);
        hipLaunchKernelGGL(calib_write4, grid(bytes / 4), dim3(256), 0, 0, (unsigned *)buf, bytes / 4);
        hipLaunchKernelGGL(calib_write1, grid(bytes), dim3(256), 0, 0, buf, bytes);
        hipLaunchKernelGGL(calib_rmw16_nt, grid(bytes / 16), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16);
        (void)hipDeviceSynchronize();
    }
    printf("done: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
