// libmvfxbench.so, measurement only -- the ceiling bench.py quotes beside the 8 TB/s spec: the in-place read-modify-write
// memory shape of hsvfilter (one 16-byte non-temporal load and store per lane, trivial arithmetic; the best shape of
// tools/probe_rmw.hip, profiles/r1/probe_rmw_steady_ceilings.txt) run over the SAME resident frame pool, in the same
// process and clock state as the timed legs.  Replaces round 2's torch copy (an out-of-place copy kernel the product
// never runs, which the product kernel outran).  The product library does not link this file.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void mvfxbench_rmw_kernel(u32x4 *buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v = NT ? __builtin_nontemporal_load(buf + i) : buf[i];
    v.x ^= 0x00010203u; v.y ^= 0x00010203u; v.z ^= 0x00010203u; v.w ^= 0x00010203u;
    if (NT) __builtin_nontemporal_store(v, buf + i); else buf[i] = v;
}

// The same shape with the stores of round 6's streaming kernels (csrc/device_store.hpp): cached 16-byte load, write-through store with the
// non-temporal hint.  (s_nop 2: the wait states an inline-asm store of more than 64 bits of data needs.)
__global__ __launch_bounds__(256) void mvfxbench_rmw_wt_kernel(u32x4 *buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v = buf[i];
    v.x ^= 0x00010203u; v.y ^= 0x00010203u; v.z ^= 0x00010203u; v.w ^= 0x00010203u;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(buf + i), "v"(v) : "memory");
}

// Out-of-place twin (16-byte load from src, 16-byte store to dst): the shape of hsvdetector / colorlut / the converters.
__global__ __launch_bounds__(256) void mvfxbench_copy_kernel(const u32x4 *src, u32x4 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v = __builtin_nontemporal_load(src + i);
    v.x ^= 0x00010203u;
    __builtin_nontemporal_store(v, dst + i);
}

extern "C" {

// `regions` regions of `bytes_per_launch` bytes (multiple of 16) starting at `buf`, visited round-robin, one launch each;
// `warm` untimed launches, then `launches` timed ones between two HIP events on `stream`.  mode 0: in place, non-temporal;
// 1: in place, cached; 2: out of place (region k -> region k+1, non-temporal); 3: in place, cached load + write-through store.  Every region is
// XOR-ed an even number of times when (warm + launches) is a multiple of 2 * regions (modes 0/1/3), i.e. the pool is left as it was.
int mvfxbench_rmw_ceiling(void *buf, size_t bytes_per_launch, uint32_t regions, uint32_t warm, uint32_t launches, int mode,
                          void *stream, double *gbs_out, double *us_per_launch_out)
{
    if (!buf || bytes_per_launch < 16 || (bytes_per_launch & 15) || regions == 0 || launches == 0 || (mode == 2 && regions < 2))
        return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = bytes_per_launch / 16;
    const unsigned grid = (unsigned)((n + 255) / 256);
    auto launch = [&](uint32_t k) {
        u32x4 *p = (u32x4 *)((char *)buf + (size_t)(k % regions) * bytes_per_launch);
        if (mode == 0) hipLaunchKernelGGL(mvfxbench_rmw_kernel<true>, dim3(grid), dim3(256), 0, st, p, n);
        else if (mode == 1) hipLaunchKernelGGL(mvfxbench_rmw_kernel<false>, dim3(grid), dim3(256), 0, st, p, n);
        else if (mode == 3) hipLaunchKernelGGL(mvfxbench_rmw_wt_kernel, dim3(grid), dim3(256), 0, st, p, n);
        else hipLaunchKernelGGL(mvfxbench_copy_kernel, dim3(grid), dim3(256), 0, st, p,
                                (u32x4 *)((char *)buf + (size_t)((k + 1) % regions) * bytes_per_launch), n);
    };
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -4;
    for (uint32_t k = 0; k < warm; k++) launch(k);
    (void)hipEventRecord(e0, st);
    for (uint32_t k = 0; k < launches; k++) launch(warm + k);
    (void)hipEventRecord(e1, st);
    const hipError_t err = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess || hipGetLastError() != hipSuccess || ms <= 0.f) return -4;
    const double per_s = (double)ms * 1e-3 / launches;
    if (us_per_launch_out) *us_per_launch_out = per_s * 1e6;
    if (gbs_out) *gbs_out = 2.0 * (double)bytes_per_launch / per_s / 1e9;   // bytes read + bytes written
    return 0;
}

} // extern "C"
