// tools/probe_rmw.hip -- steady-state ceiling of the in-place read-modify-write shape of hsvfilter (16 B per lane
// in, same 16 B out, trivial arithmetic), for the load/store variants one could pick.  531 MB per launch cycling
// through a 12.7 GB pool like bench.py; 0.5 s of untimed launches first (clock ramp).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_rmw.hip -o /tmp/probe_rmw && /tmp/probe_rmw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define TRIV(v) do { (v).x ^= 0x00010203u; (v).y ^= 0x00010203u; (v).z ^= 0x00010203u; (v).w ^= 0x00010203u; } while (0)

template <int TILE, bool NT, bool ADJ, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_rmw(u32x4 *buf, size_t n)
{
    u32x4 v[TILE];
    size_t idx[TILE];
#pragma unroll
    for (int u = 0; u < TILE; u++) {
        idx[u] = ADJ ? ((size_t)blockIdx.x * BLOCK + threadIdx.x) * TILE + u : (size_t)blockIdx.x * (BLOCK * TILE) + (size_t)u * BLOCK + threadIdx.x;
        if (idx[u] < n) v[u] = NT ? __builtin_nontemporal_load(buf + idx[u]) : buf[idx[u]];
    }
#pragma unroll
    for (int u = 0; u < TILE; u++)
        if (idx[u] < n) {
            TRIV(v[u]);
            if (NT) __builtin_nontemporal_store(v[u], buf + idx[u]); else buf[idx[u]] = v[u];
        }
}

template <int TILE, bool NT, bool ADJ, int BLOCK>
static void run(const char *name, u32x4 *pool, size_t pool_n, size_t n)
{
    const size_t launches_in_pool = pool_n / n;
    const unsigned grid = (unsigned)((n + (size_t)BLOCK * TILE - 1) / ((size_t)BLOCK * TILE));
    size_t k = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.4) {
        for (int i = 0; i < 50; i++, k++)
            hipLaunchKernelGGL((k_rmw<TILE, NT, ADJ, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, pool + (k % launches_in_pool) * n, n);
        (void)hipDeviceSynchronize();
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 1000;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++, k++)
        hipLaunchKernelGGL((k_rmw<TILE, NT, ADJ, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, pool + (k % launches_in_pool) * n, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms / reps;
    printf("%-44s %8.1f us/launch  %7.0f GB/s  (%5.1f k 4K-frames/s)\n", name, per * 1e3, 2.0 * n * 16 / (per * 1e-3) / 1e9,
           n * 16 / (3840.0 * 2160 * 4) / (per * 1e-3) / 1e3);
}

int main()
{
    const size_t frame = (size_t)3840 * 2160 * 4, n = 16 * frame / 16, pool_n = 24 * n;
    u32x4 *pool;
    if (hipMalloc(&pool, pool_n * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(pool, 7, pool_n * 16);
    run<1, false, false, 256>("1 x 16 B per lane, cached", pool, pool_n, n);
    run<1, true, false, 256>("1 x 16 B per lane, non-temporal", pool, pool_n, n);
    run<2, false, false, 256>("2 x 16 B per lane (block stride), cached", pool, pool_n, n);
    run<2, true, false, 256>("2 x 16 B per lane (block stride), nt", pool, pool_n, n);
    run<2, true, true, 256>("2 x 16 B per lane (adjacent 32 B), nt", pool, pool_n, n);
    run<4, true, false, 256>("4 x 16 B per lane (block stride), nt", pool, pool_n, n);
    run<4, true, true, 256>("4 x 16 B per lane (adjacent 64 B), nt", pool, pool_n, n);
    run<8, true, false, 256>("8 x 16 B per lane (block stride), nt", pool, pool_n, n);
    run<2, true, false, 512>("2 x 16 B per lane, nt, 512-thread groups", pool, pool_n, n);
    run<2, true, false, 1024>("2 x 16 B per lane, nt, 1024-thread groups", pool, pool_n, n);
    run<2, true, false, 128>("2 x 16 B per lane, nt, 128-thread groups", pool, pool_n, n);
    run<2, true, false, 64>("2 x 16 B per lane, nt, 64-thread groups", pool, pool_n, n);
    return 0;
}
