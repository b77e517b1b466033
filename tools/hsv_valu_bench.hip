// tools/hsv_valu_bench.hip -- VALU-only cost of the hsvfilter pixel pipeline on gfx950, with parts
// deleted one at a time ("cost attribution by deletion").  No memory traffic in the timed loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//         -Igst-plugin-rs_amd/csrc -Iinclude tools/hsv_valu_bench.hip -o /tmp/hsv_valu_bench
#include "hsv_math.hpp"

#include <cstdio>
#include <cstring>
#include <cmath>

using namespace mvfx;

enum : unsigned { NO_RCP = 1, NO_SDWA = 2, NO_LDS = 4, NO_SELECT = 8, NO_CVTIN = 16, NO_HUEWRAP = 32, NO_SATVAL = 64,
                  NO_TINY = 128, NO_MAXMIN = 256, NO_DIV255 = 512, NO_FRACT = 1024, NO_OUT255 = 2048,
                  ALT_ONE_RCP = 1u << 12, ALT_FMAC_NEG = 1u << 13, ALT_SEXT_MAGIC = 1u << 14, ALT_W_FMA = 1u << 15, ALT_KVGPR = 1u << 16 };

template <unsigned CFG>
__device__ __forceinline__ float divx(float n, float d)
{
    const float y = (CFG & NO_RCP) ? d : __builtin_amdgcn_rcpf(d);
    const float q0 = n * y;
    const float r = __builtin_fmaf(-d, q0, n);
    return fmac_vv(q0, r, y);
}

// quotient from a given reciprocal estimate y ~ 1/d (residual correction), plain and sign-flipped forms
__device__ __forceinline__ float quot(float n, float d, float y)
{
    const float q0 = n * y;
    const float r = __builtin_fmaf(-d, q0, n);
    return fmac_vv(q0, r, y);
}
// returns -RN(n/d) from yn ~ -1/d using only VOP2 fmac: q0n = n*yn = -q0; r = n + d*q0n; -q = q0n + r*yn
__device__ __forceinline__ float quot_neg(float n, float d, float yn)
{
    const float q0n = n * yn;
    const float r = fmac_vv(n, d, q0n);
    return fmac_vv(q0n, r, yn);
}
__device__ __forceinline__ float rcp_neg(float d)
{
    float y;
    asm("v_rcp_f32_e64 %0, -%1" : "=v"(y) : "v"(d));
    return y;
}

template <unsigned CFG>
__device__ __forceinline__ uint32_t pipeline(uint32_t px, const FastConsts &k, const uint32_t *lut)
{
    float f0, f1, f2;
    if (CFG & NO_CVTIN) {
        f0 = __uint_as_float(px & 0x3f8000ffu); f1 = __uint_as_float(px & 0x3f80ff00u); f2 = __uint_as_float(px & 0x3fff0000u);
    } else {
        f0 = (float)(px & 0xffu); f1 = (float)((px >> 8) & 0xffu); f2 = (float)((px >> 16) & 0xffu);
    }
    float r, g, b;
    if (CFG & NO_DIV255) { r = f0; g = f1; b = f2; } else { r = div255(f0, k); g = div255(f1, k); b = div255(f2, k); }
    float value, minv;
    if (CFG & NO_MAXMIN) { value = r + g; minv = g - b; } else { value = fmaxf(r, fmaxf(g, b)); minv = fminf(r, fminf(g, b)); }
    const float chroma = value - minv;
    const float dgb = g - b, dbr = b - r, drg = r - g;
    float n, off;
    if (CFG & NO_SELECT) { n = dgb + dbr * drg; off = 2.0f; }
    else {
        const bool is_r = (r == value), is_g = (g == value);
        n = is_r ? dgb : (is_g ? dbr : drg);
        off = is_r ? 0.0f : (is_g ? 2.0f : 4.0f);
    }
    const float d1 = (CFG & NO_TINY) ? chroma : chroma + k.tiny;
    const float d2 = (CFG & NO_TINY) ? value : value + k.tiny;
    float q, sq;
    Hsv o;
    float hue;
    if (CFG & ALT_ONE_RCP) {
        if (CFG & ALT_FMAC_NEG) {
            const float y12n = rcp_neg(d1 * d2);
            const float qn = quot_neg(n, d1, y12n * d2), sn = quot_neg(chroma, d2, y12n * d1);
            hue = (qn - off) * -k.k60; // == (off + q) * 60 (off negation is free: constants)
            sq = sn;                   // consumer multiplies by -saturation_mul
        } else {
            const float y12 = __builtin_amdgcn_rcpf(d1 * d2);
            q = quot(n, d1, y12 * d2); sq = quot(chroma, d2, y12 * d1);
            hue = (off + q) * k.k60;
        }
    } else if (CFG & ALT_FMAC_NEG) {
        const float qn = quot_neg(n, d1, rcp_neg(d1)), sn = quot_neg(chroma, d2, rcp_neg(d2));
        hue = (qn - off) * -k.k60;
        sq = sn;
    } else {
        q = divx<CFG>(n, d1);
        hue = (off + q) * k.k60;
        sq = divx<CFG>(chroma, d2);
    }
    if (CFG & NO_HUEWRAP) o.h = hue; else o.h = hue + __uint_as_float(sign_mask(hue) & k.bits360);
    o.s = sq;
    o.v = value;
    // filter
    const float x = o.h + k.hue_shift;
    if (CFG & NO_HUEWRAP) o.h = x; else o.h = x - __uint_as_float(sign_mask(k.pred360 - x) & k.bits360);
    if (CFG & NO_SATVAL) { o.s = o.s * k.saturation_mul; o.v = o.v * k.value_mul; }
    else { o.s = add_clamp01(k.saturation_mul * o.s, k.saturation_off); o.v = add_clamp01(k.value_mul * o.v, k.value_off); }
    // to_rgb
    const float c = o.v * o.s;
    float hp, a, w;
    uint32_t lds_addr = 0;
    if (CFG & ALT_SEXT_MAGIC) {
        const float hh = fmac_sv(o.h * (0.5f * k.c60lo), 0.5f * k.c60, o.h); // RN(h/120) (constants would be precomputed)
        const float t = hh + 1048575.9375f;                                   // 2^20 - 1/16: mantissa bits 2..4 = floor(2hh)
        lds_addr = __float_as_uint(t) & 28u;
        hp = hh;
        const float f = __builtin_amdgcn_fractf(hh);
        if (CFG & ALT_W_FMA) { const float gg = f - 0.5f; w = __builtin_fmaf(-2.0f, fabsf(gg), 1.0f); }
        else { a = __builtin_fmaf(f, 2.0f, -1.0f); w = 1.0f - fabsf(a); }
    } else {
        hp = div60(o.h, k);
        if (CFG & NO_FRACT) { a = hp * 0.5f; w = 1.0f - fabsf(a); }
        else {
            const float f = __builtin_amdgcn_fractf(0.5f * hp);
            if (CFG & ALT_W_FMA) { const float gg = f - 0.5f; w = __builtin_fmaf(-2.0f, fabsf(gg), 1.0f); }
            else { a = __builtin_fmaf(f, 2.0f, -1.0f); w = 1.0f - fabsf(a); }
        }
    }
    const float xx = c * w;
    const float m = o.v - c;
    float yc, yx, y0;
    if (CFG & NO_OUT255) { yc = c + m; yx = xx + m; y0 = m; } else { yc = (c + m) * k.k255; yx = (xx + m) * k.k255; y0 = m * k.k255; }
    uint32_t T;
    if (CFG & NO_SDWA) T = __float_as_uint(yc) ^ __float_as_uint(yx) ^ __float_as_uint(y0);
    else {
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(T) : "v"(yc));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(yx));
        asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(y0));
    }
    if (CFG & NO_LDS) return T ^ px ^ __float_as_uint(hp);
    if (CFG & ALT_SEXT_MAGIC)
        return __builtin_amdgcn_perm(T, px, *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lut) + lds_addr));
    const uint32_t sext = (uint32_t)__float2uint_rz(hp);
    return __builtin_amdgcn_perm(T, px, lut[sext & 7]);
}

template <unsigned CFG>
__global__ __launch_bounds__(256) void valu_kernel(uint4 *io, FastConsts k, int iters)
{
    __shared__ uint32_t lut[8];
    if (threadIdx.x < 8) lut[threadIdx.x] = sextant_selector(threadIdx.x, 0, false);
    __syncthreads();
    uint4 v = io[blockIdx.x * 256 + threadIdx.x];
    if (CFG & ALT_KVGPR) { // constants in VGPRs instead of SGPRs
        uint32_t *w = reinterpret_cast<uint32_t *>(&k);
#pragma unroll
        for (unsigned i = 0; i < sizeof(FastConsts) / 4; i++)
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "s"(w[i]));
    }
    for (int i = 0; i < iters; i++) {
        v.x = pipeline<CFG>(v.x, k, lut);
        v.y = pipeline<CFG>(v.y, k, lut);
        v.z = pipeline<CFG>(v.z, k, lut);
        v.w = pipeline<CFG>(v.w, k, lut);
    }
    io[blockIdx.x * 256 + threadIdx.x] = v;
}

static FastConsts consts()
{
    FastConsts k{};
    k.c255 = 1.0f / 255.0f; k.c255lo = (float)(1.0 / 255.0 - (double)k.c255);
    k.c60 = 1.0f / 60.0f; k.c60lo = (float)(1.0 / 60.0 - (double)k.c60);
    k.k255 = 255.0f; k.k60 = 60.0f; k.k360 = 360.0f; k.pred360 = nextafterf(360.0f, 0.0f); k.tiny = 1e-30f;
    const float f360 = 360.0f; memcpy(&k.bits360, &f360, 4);
    k.hue_shift = 90.0f; k.saturation_mul = 1.25f; k.saturation_off = -0.05f; k.value_mul = 0.9f; k.value_off = 0.02f;
    return k;
}

template <unsigned CFG>
static float run(const char *name, uint4 *buf, float base)
{
    const int blocks = 256 * 8, iters = 400;
    const FastConsts k = consts();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<CFG>, dim3(blocks), dim3(256), 0, 0, buf, k, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(valu_kernel<CFG>, dim3(blocks), dim3(256), 0, 0, buf, k, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // ns of one SIMD per 64-pixel wave-row: 8 waves/SIMD x 4 px x iters rows
    const double ns = best * 1e6 / (8.0 * 4 * iters);
    const double fps = 1024.0 * 64 / ns * 1e9 / (3840.0 * 2160);
    printf("%-28s %8.3f ms  %7.1f ns/wave-row  (4K fps if VALU-only: %6.0f)  delta %+6.1f ns\n", name, best, ns, fps, base > 0 ? ns - base : 0.0);
    return (float)ns;
}

int main()
{
    uint4 *buf; (void)hipMalloc(&buf, 2048 * 256 * 16);
    uint32_t *h = (uint32_t *)malloc(2048 * 256 * 16);
    uint32_t s = 12345;
    for (int i = 0; i < 2048 * 256 * 4; i++) { s = s * 1664525u + 1013904223u; h[i] = s; }
    (void)hipMemcpy(buf, h, 2048 * 256 * 16, hipMemcpyHostToDevice);
    const float base = run<0>("full pipeline", buf, 0);
#define R(c) run<c>(#c, buf, base)
    R(NO_RCP); R(NO_SDWA); R(NO_LDS); R(NO_SELECT); R(NO_CVTIN); R(NO_HUEWRAP); R(NO_SATVAL); R(NO_TINY); R(NO_MAXMIN);
    R(NO_DIV255); R(NO_FRACT); R(NO_OUT255);
    R(ALT_KVGPR); R(ALT_KVGPR | ALT_FMAC_NEG | ALT_SEXT_MAGIC | ALT_W_FMA);
    R(ALT_ONE_RCP); R(ALT_FMAC_NEG); R(ALT_ONE_RCP | ALT_FMAC_NEG); R(ALT_SEXT_MAGIC); R(ALT_W_FMA); R(ALT_SEXT_MAGIC | ALT_W_FMA);
    R(ALT_ONE_RCP | ALT_FMAC_NEG | ALT_SEXT_MAGIC | ALT_W_FMA);
    R(NO_RCP | NO_SDWA | NO_LDS | NO_SELECT);
    R(NO_RCP | NO_SDWA | NO_LDS | NO_SELECT | NO_CVTIN | NO_MAXMIN | NO_FRACT);
    return 0;
}
