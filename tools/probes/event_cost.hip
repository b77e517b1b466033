// What a fence costs the device: a 4K-frame-sized read-modify-write kernel launched back to back on two alternating streams with
// (a) nothing behind it, (b) hipEventRecord behind every launch, (c) the event attached to the launch itself as the stop event of
// hipExtLaunchKernelGGL (the dispatch packet's own completion signal: no barrier packet).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rmw(u32x4 *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        u32x4 v = __builtin_nontemporal_load(p + i);
        v.x += 1; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
        __builtin_nontemporal_store(v, p + i);
    }
}

int main()
{
    const size_t bytes = 3840ull * 2160 * 4, n = bytes / 16;
    const int pool = 8, launches = 6000;
    std::vector<u32x4 *> bufs(pool);
    for (auto &b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
    hipStream_t st[2];
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<hipEvent_t> evs(64);
    for (auto &e : evs) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (int mode = 0; mode < 4; mode++) {
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < launches; k++) {
                hipStream_t s = st[k & 1];
                hipEvent_t e = evs[k & 63];
                if (mode == 2) hipExtLaunchKernelGGL(rmw, grid, block, 0, s, nullptr, e, 0, bufs[k % pool], n);
                else hipLaunchKernelGGL(rmw, grid, block, 0, s, bufs[k % pool], n);
                if (mode == 1) hipEventRecord(e, s);
                if (mode == 3) { hipEventRecord(e, s); hipEventRecord(evs[(k + 32) & 63], s); }
            }
            hipDeviceSynchronize();
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::max(best, launches / secs);
        }
        const char *names[] = {"no fence", "hipEventRecord behind every launch", "event attached as the launch's stop event (hipExtLaunchKernelGGL)",
                               "two hipEventRecord behind every launch"};
        printf("%-70s %8.0f launches/s  (%.2f us per launch, %.3f of 8 TB/s)\n", names[mode], best, 1e6 / best, best * 2 * bytes / 8e12);
    }
    // does the attached event order other streams and the host?
    hipEvent_t e = evs[0];
    hipExtLaunchKernelGGL(rmw, grid, block, 0, st[0], nullptr, e, 0, bufs[0], n);
    const hipError_t q0 = hipEventQuery(e);
    hipStreamWaitEvent(st[1], e, 0);
    hipEventSynchronize(e);
    printf("attached event: query right behind the launch -> %s, after synchronize -> %s\n", hipGetErrorName(q0), hipGetErrorName(hipEventQuery(e)));
    return 0;
}
