// tools/probes/typed_unaligned.hip -- does buffer_load_format_xyz (DATA_FORMAT 8_8_8_8, UNORM) work at byte offsets that are NOT multiples
// of four?  If it does, a packed RGB pixel (3 bytes at offset 3 i) can be fetched as three exact RN(byte / 255) floats by the texture
// unit, like the RGBA kernels do, and the 3-byte formats lose their unpacking instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/typed_unaligned.hip -o tools/probes/typed_unaligned.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t *p, float *o, uint32_t bytes, uint32_t word3, uint32_t step)
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xffff); // stride 0: raw byte offsets
    rs.z = __builtin_amdgcn_readfirstlane((int)bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)word3);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, off = i * step;
    f4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rs) : "memory");
    float *d = o + (size_t)i * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

int main()
{
    const int n = 256;
    uint8_t h[n * 4 + 8];
    uint64_t x = 0x5EED0001ull;
    for (size_t i = 0; i < sizeof h; i++) { x = x * 6364136223846793005ull + 1442695040888963407ull; h[i] = (uint8_t)(x >> 56); }
    uint8_t *d; float *o;
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&o, n * 16) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (uint32_t step : {4u, 3u, 1u, 2u}) {
        (void)hipMemset(o, 0xff, n * 16);
        hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, o, (uint32_t)sizeof h, 0x00050FACu, step);
        if (hipDeviceSynchronize() != hipSuccess) { printf("step %u: launch failed\n", step); continue; }
        float r[n * 4];
        (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int bad = 0, aligned_down = 0;
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 4; c++) {
                const float want = (float)h[i * step + c] / 255.0f, down = (float)h[((i * step) & ~3u) + c] / 255.0f;
                if (r[i * 4 + c] != want) { bad++; if (r[i * 4 + c] == down) aligned_down++; }
            }
        printf("byte offset = %u x lane: %d of %d channel values differ from RN(byte / 255) at the UNALIGNED address (%d of those equal the value at "
               "the address rounded down to 4)\n", step, bad, n * 4, aligned_down);
    }
    return 0;
}
