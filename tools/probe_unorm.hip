// tools/probe_unorm.hip -- does a typed buffer load (buffer_load_format_xyzw, DATA_FORMAT 8_8_8_8, NUM_FORMAT UNORM)
// return RN(b / 255.0f) for every byte value?  If it does, the texture unit does hsvfilter's / hsvdetector's /
// colorlut's three `u8 / 255` divisions (9 VALU instructions per pixel) for free.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_unorm.hip -o tools/probe_unorm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t *p, float *o, uint32_t bytes, uint32_t word3)
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xffff); // stride 0
    rs.z = __builtin_amdgcn_readfirstlane((int)bytes);
    rs.w = __builtin_amdgcn_readfirstlane((int)word3);
    const uint32_t off = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    f4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rs) : "memory");
    float *d = o + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

int main()
{
    const int n = 256; // pixel i = (i, 255-i, i^0x55, (i*7)&255)
    uint8_t h[n * 4];
    for (int i = 0; i < n; i++) { h[4 * i] = i; h[4 * i + 1] = 255 - i; h[4 * i + 2] = i ^ 0x55; h[4 * i + 3] = (i * 7) & 255; }
    uint8_t *d; float *o;
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&o, n * 16) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (uint32_t word3 : {0x00050FACu, 0x00050FACu | (1u << 24)}) {
        (void)hipMemset(o, 0xff, n * 16);
        hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, o, (uint32_t)sizeof h, word3);
        if (hipDeviceSynchronize() != hipSuccess) { printf("word3 %08x: launch failed\n", word3); continue; }
        float r[n * 4];
        (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int bad = 0, ulp1 = 0;
        for (int i = 0; i < n * 4; i++) {
            const float want = (float)h[i] / 255.0f; // IEEE division, RN
            if (r[i] != want) {
                bad++;
                uint32_t a, b; memcpy(&a, &r[i], 4); memcpy(&b, &want, 4);
                if (a + 1 == b || b + 1 == a) ulp1++;
                if (bad <= 6) printf("  byte %3u: got %.9g (%08x) want %.9g (%08x)\n", h[i], r[i], a, want, b);
            }
        }
        printf("word3 %08x: %d of %d values differ from RN(b/255) (%d of them by 1 ulp); sample: %g %g %g %g\n", word3, bad, n * 4, ulp1, r[4], r[5], r[6], r[7]);
    }
    return 0;
}
