// Per-pixel RGB<->HSV arithmetic of the hsv plugin, written for gfx950.
//
// Two variants, both bit-exact with the reference (video/hsv/src/hsvutils.rs) and selected by
// the host (hsv_kernels.hip):
//
//  GENERAL  literal transcription: IEEE f32 divides (the compiler's correctly rounded
//           v_div_scale/v_rcp/v_fma/v_div_fmas/v_div_fixup sequence) and ocml fmodf.  Valid for
//           every settings vector including NaN/inf/huge hue shifts.  ~250 VALU ops per pixel
//           => VALU-bound at roughly a quarter of the HBM roofline.
//
//  FAST     the same values through exact strength reductions, each proven exhaustively:
//           - u8/255, h/60 as mul+fma+fma          (tools/prove_exact.c P1, P2; Markstein)
//           - fmod(h',2) as 2*fract(h'/2)           (P3)
//           - `hue % 360` and the [0,1] clamps in from_rgb dropped (P4: identities on all 2^24
//             inputs); epsilon compares == "channel is the max" (P5)
//           - hue wrap as conditional +-360         (P7), needs |hue_shift| <= 360
//           - (g-b)/chroma and chroma/value as v_rcp_f32 + one Newton step on the quotient;
//             proven on the GPU by the exhaustive 2^24-triple parity test
//             (tests/test_hsv_gpu.py::test_from_rgb_f32_exhaustive), since both divides depend
//             only on (R,G,B), never on the settings.
//           Domain (checked on the host, hsv_kernels.hip fast_domain_ok): all five settings
//           finite, |hue_shift| <= 360 and hue_shift == 0 or |hue_shift| >= 1e-30 (keeps h/60
//           out of the denormal range where P2/P3 do not hold).
//
// Built with -ffp-contract=off: every a*b+c below is two roundings unless written as
// __builtin_fmaf (reference never fuses).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mvfx {

struct HsvFilterParams { // hsvfilter/imp.rs:32-39
    float hue_shift, saturation_mul, saturation_off, value_mul, value_off;
};

struct HsvDetectorParams { // hsvdetector/imp.rs:34-42, plus ref_hue_offset = 180 - hue_ref (:141)
    float ref_hue_offset, hue_var, saturation_ref, saturation_var, value_ref, value_var;
};

struct Hsv {
    float h, s, v;
};

constexpr int kGeneral = 0;
constexpr int kFast = 1;

// hsvutils.rs:16-38 custom Clamp: self.max(lo).min(hi), NaN-ignoring => clamp(NaN) == lo
__device__ __forceinline__ float hsv_clamp(float v, float lo, float hi)
{
    return fminf(fmaxf(v, lo), hi);
}

// ---------------------------------------------------------------- GENERAL (literal)

// hsvutils.rs:44-84 / :88-128 on the true (R,G,B)
__device__ __forceinline__ Hsv from_rgb_general(uint32_t R, uint32_t G, uint32_t B)
{
    const float r = (float)R / 255.0f;
    const float g = (float)G / 255.0f;
    const float b = (float)B / 255.0f;
    const uint32_t mx = max(R, max(G, B));
    const uint32_t mn = min(R, min(G, B));
    const float value = (float)mx / 255.0f;
    const float chroma = value - ((float)mn / 255.0f);

    float hue;
    if (chroma == 0.0f) {
        hue = 0.0f;
    } else if (fabsf(value - r) < 0.00001f) {
        hue = 60.0f * ((g - b) / chroma);
    } else if (fabsf(value - g) < 0.00001f) {
        hue = 60.0f * (2.0f + ((b - r) / chroma));
    } else if (fabsf(value - b) < 0.00001f) {
        hue = 60.0f * (4.0f + ((r - g) / chroma));
    } else {
        hue = 0.0f;
    }
    if (hue < 0.0f)
        hue += 360.0f;
    const float saturation = (value == 0.0f) ? 0.0f : chroma / value;

    Hsv o;
    o.h = fmodf(hue, 360.0f);
    o.s = hsv_clamp(saturation, 0.0f, 1.0f);
    o.v = hsv_clamp(value, 0.0f, 1.0f);
    return o;
}

// `as u8` on an f32 that was clamped to [0,255] by hsv_clamp (NaN already mapped to 0)
__device__ __forceinline__ uint32_t trunc_u8(float v) { return (uint32_t)__float2uint_rz(v); }

// hsvutils.rs:132-163 / :167-198; returns the (R,G,B) bytes
__device__ __forceinline__ void to_rgb_general(const Hsv in, uint32_t &R, uint32_t &G, uint32_t &B)
{
    const float c = in.v * in.s;
    const float hue_prime = in.h / 60.0f;
    const float x = c * (1.0f - fabsf(fmodf(hue_prime, 2.0f) - 1.0f));

    float p0, p1, p2;
    if (hue_prime < 0.0f) {
        p0 = 0.0f; p1 = 0.0f; p2 = 0.0f;
    } else if (hue_prime <= 1.0f) {
        p0 = c; p1 = x; p2 = 0.0f;
    } else if (hue_prime <= 2.0f) {
        p0 = x; p1 = c; p2 = 0.0f;
    } else if (hue_prime <= 3.0f) {
        p0 = 0.0f; p1 = c; p2 = x;
    } else if (hue_prime <= 4.0f) {
        p0 = 0.0f; p1 = x; p2 = c;
    } else if (hue_prime <= 5.0f) {
        p0 = x; p1 = 0.0f; p2 = c;
    } else if (hue_prime <= 6.0f) {
        p0 = c; p1 = 0.0f; p2 = x;
    } else {
        p0 = 0.0f; p1 = 0.0f; p2 = 0.0f;
    }
    const float m = in.v - c;
    R = trunc_u8(hsv_clamp((p0 + m) * 255.0f, 0.0f, 255.0f));
    G = trunc_u8(hsv_clamp((p1 + m) * 255.0f, 0.0f, 255.0f));
    B = trunc_u8(hsv_clamp((p2 + m) * 255.0f, 0.0f, 255.0f));
}

// hsvfilter/imp.rs:102-115
__device__ __forceinline__ Hsv filter_hsv_general(Hsv hsv, const HsvFilterParams &p)
{
    hsv.h = fmodf(hsv.h + p.hue_shift, 360.0f);
    if (hsv.h < 0.0f)
        hsv.h += 360.0f;
    hsv.s = hsv_clamp(p.saturation_mul * hsv.s + p.saturation_off, 0.0f, 1.0f);
    hsv.v = hsv_clamp(p.value_mul * hsv.v + p.value_off, 0.0f, 1.0f);
    return hsv;
}

// hsvdetector/imp.rs:141-155 -> 255 / 0
__device__ __forceinline__ uint32_t detect_alpha_general(const Hsv hsv, const HsvDetectorParams &p)
{
    float shifted_hue = hsv.h + p.ref_hue_offset;
    if (shifted_hue < 0.0f)
        shifted_hue += 360.0f;
    shifted_hue = fmodf(shifted_hue, 360.0f);
    const bool hit = fabsf(shifted_hue - 180.0f) <= p.hue_var &&
                     fabsf(hsv.s - p.saturation_ref) <= p.saturation_var &&
                     fabsf(hsv.v - p.value_ref) <= p.value_var;
    return hit ? 255u : 0u;
}

// ---------------------------------------------------------------- FAST (exact reductions)

// RN(x/255) for x an integer-valued float in [0,255]  (prove_exact P1)
__device__ __forceinline__ float div255(float x)
{
    const float c = 1.0f / 255.0f;
    const float q0 = x * c;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q0, x), c, q0);
}

// RN(h/60) for h == 0 or h in [1e-30,360]  (prove_exact P2)
__device__ __forceinline__ float div60(float h)
{
    const float c = 1.0f / 60.0f;
    const float q0 = h * c;
    return __builtin_fmaf(__builtin_fmaf(-60.0f, q0, h), c, q0);
}

// RN(n/d) for the two from_rgb quotients, d in [1/255,1] (never 0 here), |n| <= d.
// v_rcp_f32 (<= 1 ulp) seeds one residual correction of the quotient.  Correct rounding on all
// 2^24 (R,G,B) is established by the exhaustive GPU parity test, not by analysis.
__device__ __forceinline__ float div_rgb(float n, float d)
{
    const float y = __builtin_amdgcn_rcpf(d);
    const float q0 = n * y;
    const float r = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(r, y, q0);
}

__device__ __forceinline__ Hsv from_rgb_fast(uint32_t R, uint32_t G, uint32_t B)
{
    const float r = div255((float)R);
    const float g = div255((float)G);
    const float b = div255((float)B);
    // RN is monotone, so max/min of the quotients == quotient of the max/min byte
    const float value = fmaxf(r, fmaxf(g, b));
    const float minv = fminf(r, fminf(g, b));
    const float chroma = value - minv;

    // branch ladder of hsvutils.rs:61-71 as selects (P5: eps test == "is the max")
    const bool is_r = (r == value);
    const bool is_g = (g == value);
    const float n1 = is_r ? g : (is_g ? b : r);
    const float n2 = is_r ? b : (is_g ? r : g);
    const float off = is_r ? 0.0f : (is_g ? 2.0f : 4.0f);
    // chroma == 0 => all channels equal => n1 - n2 == 0, is_r => hue = 60*(0+0) = 0 as required;
    // the denominator only has to be non-zero there.
    const float q = div_rgb(n1 - n2, fmaxf(chroma, 1e-30f));
    float hue = 60.0f * (off + q); // off == 0: q + 0 is exact, matches the un-added branch
    const float wrapped = hue + 360.0f;
    hue = (hue < 0.0f) ? wrapped : hue;

    Hsv o;
    o.h = hue;                                         // P4: hue % 360 == hue
    o.s = div_rgb(chroma, fmaxf(value, 1e-30f));      // value == 0 => chroma == 0 => 0
    o.v = value;                                       // P4: clamps are identities
    return o;
}

// hsvfilter/imp.rs:102-115 for finite settings with |hue_shift| <= 360  (P7)
__device__ __forceinline__ Hsv filter_hsv_fast(Hsv hsv, const HsvFilterParams &p)
{
    const float x = hsv.h + p.hue_shift;
    const float t = (x >= 360.0f) ? x - 360.0f : x;
    hsv.h = (t < 0.0f) ? t + 360.0f : t;
    // finite settings => no NaN => med3 == max(.,0).min(1)
    hsv.s = __builtin_amdgcn_fmed3f(p.saturation_mul * hsv.s + p.saturation_off, 0.0f, 1.0f);
    hsv.v = __builtin_amdgcn_fmed3f(p.value_mul * hsv.v + p.value_off, 0.0f, 1.0f);
    return hsv;
}

// hsvutils.rs:132-163 for h in {0} U [1e-30,360], s,v in [0,1]
__device__ __forceinline__ void to_rgb_fast(const Hsv in, uint32_t &R, uint32_t &G, uint32_t &B)
{
    const float c = in.v * in.s;
    const float hp = div60(in.h);
    const float f = __builtin_amdgcn_fractf(0.5f * hp);          // P3
    const float a = __builtin_fmaf(f, 2.0f, -1.0f);              // RN(fmod(hp,2) - 1), 2f exact
    const float x = c * (1.0f - fabsf(a));

    // hp in [0,6]: the `< 0` and `> 6` arms of the ladder are unreachable
    const bool le1 = hp <= 1.0f, le2 = hp <= 2.0f, le3 = hp <= 3.0f, le4 = hp <= 4.0f,
               le5 = hp <= 5.0f;
    const float p0 = le1 ? c : le2 ? x : le4 ? 0.0f : le5 ? x : c;
    const float p1 = le1 ? x : le3 ? c : le4 ? x : 0.0f;
    const float p2 = le2 ? 0.0f : le3 ? x : le5 ? c : x;

    const float m = in.v - c;
    R = trunc_u8(__builtin_amdgcn_fmed3f((p0 + m) * 255.0f, 0.0f, 255.0f));
    G = trunc_u8(__builtin_amdgcn_fmed3f((p1 + m) * 255.0f, 0.0f, 255.0f));
    B = trunc_u8(__builtin_amdgcn_fmed3f((p2 + m) * 255.0f, 0.0f, 255.0f));
}

// ---------------------------------------------------------------- dispatch helpers

template <int VARIANT>
__device__ __forceinline__ Hsv from_rgb(uint32_t R, uint32_t G, uint32_t B)
{
    if constexpr (VARIANT == kFast)
        return from_rgb_fast(R, G, B);
    else
        return from_rgb_general(R, G, B);
}

template <int VARIANT>
__device__ __forceinline__ void hsvfilter_pixel(uint32_t &R, uint32_t &G, uint32_t &B,
                                                const HsvFilterParams &p)
{
    if constexpr (VARIANT == kFast) {
        to_rgb_fast(filter_hsv_fast(from_rgb_fast(R, G, B), p), R, G, B);
    } else {
        to_rgb_general(filter_hsv_general(from_rgb_general(R, G, B), p), R, G, B);
    }
}

} // namespace mvfx
