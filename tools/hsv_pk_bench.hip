// tools/hsv_pk_bench.hip -- VALU-only comparison of the shipped hsvfilter pixel function with a variant that
// pairs independent f32 operations into v_pk_{mul,add,fma}_f32 (two divides, two affine clamps, two /255, ...).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Igst-plugin-rs_amd/csrc -Iinclude \
//         tools/hsv_pk_bench.hip -o /tmp/hsv_pk_bench
#include "hsv_math.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace mvfx;
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_add_clamp(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct PkConsts { // the pairs a packed instruction needs side by side
    f2 neg_sat_val_mul; // {-saturation_mul, value_mul}
    f2 sat_val_off;     // {saturation_off, value_off}
};

__device__ __forceinline__ uint32_t pixel_scalar(uint32_t px, const FastConsts &k, const uint32_t *lut)
{
    const float c0 = div255((float)(px & 0xffu), k), c1 = div255((float)((px >> 8) & 0xffu), k), c2 = div255((float)((px >> 16) & 0xffu), k);
    uint32_t T;
    const uint32_t so = hsvfilter_fast_unit<false>(c0, c1, c2, k, T);
    return __builtin_amdgcn_perm(T, px, sextant_at(lut, so));
}

__device__ __forceinline__ uint32_t pixel_packed(uint32_t px, const FastConsts &k, const PkConsts &pk, const uint32_t *lut)
{
    const f2 rgf = {(float)(px & 0xffu), (float)((px >> 8) & 0xffu)};
    const f2 rg = pk_fma(rgf, (f2){k.c255, k.c255}, rgf * k.c255lo);
    const float r = rg.x, g = rg.y, b = div255((float)((px >> 16) & 0xffu), k);
    const float value = fmaxf(r, fmaxf(g, b)), minv = fminf(r, fminf(g, b));
    const float chroma = value - minv;
    const bool is_r = (r == value), is_g = (g == value);
    const float dgb = g - b, dbr = b - r, drg = r - g;
    const float n = is_r ? dgb : (is_g ? dbr : drg);
    const float off = is_r ? 0.0f : (is_g ? 2.0f : 4.0f);
    const f2 d = (f2){chroma, value} + k.tiny;
    f2 yn;
    asm("v_rcp_f32_e64 %0, -%1" : "=v"(yn.x) : "v"(d.x));
    asm("v_rcp_f32_e64 %0, -%1" : "=v"(yn.y) : "v"(d.y));
    const f2 num = {n, chroma};
    const f2 q0n = num * yn;
    const f2 res = pk_fma(d, q0n, num);
    const f2 qn = pk_fma(res, yn, q0n);                // {-q, -s}
    const float hue = (qn.x - off) * k.neg_k60;
    const float h1 = wrap_up(hue, k);
    const float h2 = wrap_down(h1 + k.hue_shift, k);
    const f2 sv = pk_add_clamp((f2){qn.y, value} * pk.neg_sat_val_mul, pk.sat_val_off); // {s', v'}
    const float c = sv.y * sv.x;
    const float hh = fmac_sv(h2 * k.c120lo, k.c120, h2);
    const float f = __builtin_amdgcn_fractf(hh);
    const uint32_t sel_off = __float_as_uint(hh + k.sext_magic) & 28u;
    const float x = c * __builtin_fmaf(-2.0f, fabsf(f - 0.5f), 1.0f);
    const float m = sv.y - c;
    const f2 y2 = ((f2){c, x} + m) * k.k255;
    const float y0 = m * k.k255;
    uint32_t T;
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(T) : "v"(y2.x));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(y2.y));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(y0));
    return __builtin_amdgcn_perm(T, px, sextant_at(lut, sel_off));
}

template <bool PACKED>
__global__ __launch_bounds__(256) void valu_kernel(uint4 *io, FastConsts k, PkConsts pk, int iters, int *mismatch)
{
    __shared__ uint32_t lut[8];
    if (threadIdx.x < 8) lut[threadIdx.x] = sextant_selector(threadIdx.x, 0, false);
    __syncthreads();
    uint4 v = io[blockIdx.x * 256 + threadIdx.x];
#ifdef MVFX_KCONST_VGPR // constants in VGPRs (experiment)
    {
        uint32_t *w = reinterpret_cast<uint32_t *>(&k);
#pragma unroll
        for (unsigned i = 0; i < sizeof(FastConsts) / 4; i++)
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "s"(w[i]));
    }
#endif
    if (iters < 0) { // parity check of the packed variant against the shipped one on this thread's pixels
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; i++)
            if (pixel_scalar(w[i], k, lut) != pixel_packed(w[i], k, pk, lut)) atomicAdd(mismatch, 1);
        return;
    }
    for (int i = 0; i < iters; i++) {
        if (PACKED) {
            v.x = pixel_packed(v.x, k, pk, lut); v.y = pixel_packed(v.y, k, pk, lut);
            v.z = pixel_packed(v.z, k, pk, lut); v.w = pixel_packed(v.w, k, pk, lut);
        } else {
            v.x = pixel_scalar(v.x, k, lut); v.y = pixel_scalar(v.y, k, lut);
            v.z = pixel_scalar(v.z, k, lut); v.w = pixel_scalar(v.w, k, lut);
        }
    }
    io[blockIdx.x * 256 + threadIdx.x] = v;
}

static FastConsts consts()
{
    FastConsts k{};
    k.c255 = 1.0f / 255.0f; k.c255lo = (float)(1.0 / 255.0 - (double)k.c255);
    k.c60 = 1.0f / 60.0f; k.c60lo = (float)(1.0 / 60.0 - (double)k.c60);
    k.c120 = 0.5f * k.c60; k.c120lo = 0.5f * k.c60lo; k.sext_magic = 1048575.9375f;
    k.k255 = 255.0f; k.k60 = 60.0f; k.neg_k60 = -60.0f; k.k360 = 360.0f; k.pred360 = nextafterf(360.0f, 0.0f); k.tiny = 1e-30f;
    const float f360 = 360.0f; memcpy(&k.bits360, &f360, 4);
    k.hue_shift = 90.0f; k.saturation_mul = 1.25f; k.saturation_off = -0.05f; k.value_mul = 0.9f; k.value_off = 0.02f;
    k.neg_saturation_mul = -1.25f;
    return k;
}

template <bool PACKED>
static void run(const char *name, uint4 *buf, const FastConsts &k, const PkConsts &pk, int *dm)
{
    const int blocks = 256 * 8, iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<PACKED>, dim3(blocks), dim3(256), 0, 0, buf, k, pk, 10, dm);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(valu_kernel<PACKED>, dim3(blocks), dim3(256), 0, 0, buf, k, pk, iters, dm);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / (8.0 * 4 * iters);
    printf("%-10s %8.3f ms  %7.1f ns/wave-row  (4K fps if VALU-only: %6.0f)\n", name, best, ns, 1024.0 * 64 / ns * 1e9 / (3840.0 * 2160));
}

int main()
{
    uint4 *buf; int *dm;
    (void)hipMalloc(&buf, 2048 * 256 * 16); (void)hipMalloc(&dm, 4); (void)hipMemset(dm, 0, 4);
    uint32_t *h = (uint32_t *)malloc(2048 * 256 * 16);
    uint32_t s = 12345;
    for (int i = 0; i < 2048 * 256 * 4; i++) { s = s * 1664525u + 1013904223u; h[i] = s; }
    (void)hipMemcpy(buf, h, 2048 * 256 * 16, hipMemcpyHostToDevice);
    const FastConsts k = consts();
    const PkConsts pk{{k.neg_saturation_mul, k.value_mul}, {k.saturation_off, k.value_off}};
    hipLaunchKernelGGL(valu_kernel<false>, dim3(2048), dim3(256), 0, 0, buf, k, pk, -1, dm);
    int mism = -1; (void)hipMemcpy(&mism, dm, 4, hipMemcpyDeviceToHost);
    printf("packed vs shipped on 2 M random pixels: %d mismatches\n", mism);
    run<false>("shipped", buf, k, pk, dm);
    run<true>("packed", buf, k, pk, dm);
    return 0;
}
