// tools/probes/aql_scope_kernel.hip -- the memory shape of the headline kernel (two 16-byte non-temporal loads per lane, a trivial operation, two
// non-temporal stores, in place) for tools/probes/aql_scope.cpp.  No blockDim / gridDim (implicit kernel arguments): 256 lanes per workgroup, fixed.
//   hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 tools/probes/aql_scope_kernel.hip -o tools/probes/aql_scope_kernel.hsaco   (a bare ELF: the HSA loader does not take clang offload bundles)
#include <hip/hip_runtime.h>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern "C" __global__ __launch_bounds__(256) void rmw_kernel(uint8_t *base, uint32_t n_groups, uint32_t key)
{
    const uint32_t g0 = blockIdx.x * 512u + threadIdx.x, g1 = g0 + 256u;
    u32x4 *p0 = reinterpret_cast<u32x4 *>(base) + g0, *p1 = reinterpret_cast<u32x4 *>(base) + g1;
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (g0 < n_groups) a = __builtin_nontemporal_load(p0);
    if (g1 < n_groups) b = __builtin_nontemporal_load(p1);
    a ^= key;
    b ^= key;
    if (g0 < n_groups) __builtin_nontemporal_store(a, p0);
    if (g1 < n_groups) __builtin_nontemporal_store(b, p1);
}
