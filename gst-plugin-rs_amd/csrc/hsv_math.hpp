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
//           - u8/255, h/120 as mul + fmac          (tools/prove_exact.c P8, P11)
//           - fmod(h',2) as 2*fract(h'/2)           (P3); 1-|2f-1| as fma(-2,|f-1/2|,1) (P14)
//           - `hue % 360` and the [0,1] clamps in from_rgb dropped (P4: identities on all 2^24
//             inputs); epsilon compares == "channel is the max" (P5)
//           - hue wraps as add + unsigned min       (P9, P13), needs |hue_shift| <= 360
//           - the sextant index out of a float's mantissa (P12)
//           - (g-b)/chroma and chroma/value as v_rcp_f32 + one residual step, carried negated (quot_neg);
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

struct FastConsts {
    float c255, c255lo;   // 1/255 = c255 + c255lo   (prove_exact P8)
    float c60, c60lo;     // 1/60  = c60 + c60lo     (P8)
    float c120, c120lo;   // halves of the above: RN(h/120) == 0.5*RN(h/60) (P11)
    float sext_magic;     // 2^20 - 1/16 (P12)
    float k255, k60, k360;
    float neg_k60;        // -60: the hue quotient is carried negated (see quot_neg)
    float pred360;        // largest float below 360 (P9)
    float tiny;           // 1e-30: keeps rcp() away from 0 without changing any non-zero value
    uint32_t bits360;     // bit pattern of 360.0f
    // hsvfilter settings (hsvfilter/imp.rs:32-39)
    float hue_shift, saturation_mul, saturation_off, value_mul, value_off;
    float neg_saturation_mul; // -saturation_mul: from_rgb hands over -s
};

struct HsvDetectorParams { // hsvdetector/imp.rs:34-42, plus ref_hue_offset = 180 - hue_ref (:141)
    float ref_hue_offset, hue_var, saturation_ref, saturation_var, value_ref, value_var;
    float k180;
    FastConsts consts;
};

struct Hsv {
    float h, s, v;
};

struct HsvN { // FAST path: saturation carried NEGATED (ns == -s), see quot_neg()
    float h, ns, v;
};

constexpr int kGeneral = 0;
constexpr int kFast = 1;    // strength-reduced, 0 <= hue_shift <= 360
constexpr int kFastNeg = 2; // strength-reduced, -360 <= hue_shift < 0
constexpr int kDetFast = 3; // hsvdetector: strength-reduced from_rgb AND hue test (finite settings, |180 - hue_ref| <= 360)

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
//
// What the instruction mix costs on MI355X (tools/probe_isa2..4.hip, tools/hsv_valu_bench.hip,
// profiles/r1/probe_isa*.txt, hsv_valu_bench_deletions.txt):
//   * in a homogeneous stream VOP1/VOP2 e32 mul/add/sub/fmac/and/ashr issue at ~2.4 cycles per wave64,
//     cvt/cmp/cndmask/max/min/fract and anything with an SGPR source at ~3.9, three-VGPR VOP3
//     (v_fma, v_perm, v_max3) ~3.7, SDWA with a preserved destination ~4.6, v_rcp_f32 ~7.6;
//   * in the real mix the pixel function averages 3.4 cycles per instruction (rocprofv3: SQ_INSTS_VALU x
//     cycles), and removing instructions pays ~1.5 % each while changing their class (SGPR -> VGPR
//     operands, tested) pays nothing: the kernel is bound by instruction COUNT at the sustained clock.
// Hence: every step below is there to delete instructions -- a*b+c as v_fmac_f32 where one operand
// dies, the two divides carried negated so that their correction steps are plain v_fmac (quot_neg),
// hue wraps as add + v_min_u32, the [0,1] clamps on the VOP3 clamp bit, the sextant read out of a
// float's mantissa, and the 6-way select of to_rgb as one v_perm_b32 through an 8-entry LDS table.

#ifdef MVFX_KCONST_VGPR // experiment knob (tools/hsv_valu_bench.hip): kernel constants held in VGPRs
#define MVFX_KC "v"
#else
#define MVFX_KC "s"
#endif
__device__ __forceinline__ float fmac_sv(float acc, float s, float v) // acc + s*v, s a kernel constant
{
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : MVFX_KC(s), "v"(v));
    return acc;
}

__device__ __forceinline__ float fmac_vv(float acc, float a, float b) // acc + a*b (one rounding)
{
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

__device__ __forceinline__ uint32_t sign_mask(float x) // 0xffffffff if the sign bit is set, else 0
{
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ float add_clamp01(float a, float s) // clamp(a + s, 0, 1) in one VOP3
{
    float r;
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), MVFX_KC(s));
    return r;
}

// `if x < 0 { x += 360 }` for x in [-360,360) \ {-0}: a negative x has the larger unsigned bit pattern,
// a non-negative x the smaller one than x + 360  (P13): v_add_f32 + v_min_u32
__device__ __forceinline__ float wrap_up(float x, const FastConsts &k)
{
    const float y = x + k.k360;
    return __uint_as_float(min(__float_as_uint(x), __float_as_uint(y)));
}
// fmod(x, 360) for x in [0,720): x - 360 is exact there, negative (larger pattern) iff x < 360  (P13)
__device__ __forceinline__ float wrap_down(float x, const FastConsts &k)
{
    const float y = x - k.k360;
    return __uint_as_float(min(__float_as_uint(y), __float_as_uint(x)));
}

// RN(x/255) for integer-valued x in [0,255]: x*C + RN(x*Clo)  (P8)
__device__ __forceinline__ float div255(float x, const FastConsts &k) { return fmac_sv(x * k.c255lo, k.c255, x); }
// RN(h/60) for h == 0 or h in [1e-30,360]  (P8)
__device__ __forceinline__ float div60(float h, const FastConsts &k) { return fmac_sv(h * k.c60lo, k.c60, h); }

// -RN(n/d) for the two from_rgb quotients, d in [1/255,1] (or `tiny`), |n| <= d.
// v_rcp_f32 (<= 1 ulp) of -d seeds one residual correction of the quotient; carrying the NEGATED
// quotient lets both correction steps be VOP2 v_fmac_f32 (no VOP3 neg modifier, measured 16 % of
// the pixel time, tools/hsv_valu_bench.hip):  yn = -1/d,  q0n = n*yn = -q0,
// r = n + d*q0n = n - d*q0 (same residual),  -q = q0n + r*yn.  Every step is the exact mirror image
// of q0 = n*y, r = fma(-d,q0,n), q = q0 + r*y, so the result is -RN(n/d) whenever that was RN(n/d).
// Correct rounding on all 2^24 (R,G,B) is established by the exhaustive GPU parity test
// (tests/test_hsv_gpu.py::test_from_rgb_f32_exhaustive), not by analysis.
__device__ __forceinline__ float quot_neg(float n, float d)
{
    float yn;
    asm("v_rcp_f32_e64 %0, -%1" : "=v"(yn) : "v"(d));
    const float q0n = n * yn;
    const float r = fmac_vv(n, d, q0n);
    return fmac_vv(q0n, r, yn);
}

// hsvutils.rs:44-84 given r,g,b = RN(byte/255) (from div255() or from the 256-entry LDS table)
__device__ __forceinline__ HsvN from_unit_rgb_fast_n(float r, float g, float b, const FastConsts &k)
{
    // RN is monotone, so max/min of the quotients == quotient of the max/min byte
    const float value = fmaxf(r, fmaxf(g, b));
    const float minv = fminf(r, fminf(g, b));
    const float chroma = value - minv;

    // branch ladder of hsvutils.rs:61-71 as selects (P5: eps test == "is the max"); the three
    // candidate numerators are cheap e32 subtractions, so only the numerator and the sextant
    // offset are selected (4 v_cndmask instead of 6)
    const bool is_r = (r == value);
    const bool is_g = (g == value);
    const float dgb = g - b, dbr = b - r, drg = r - g;
    const float n = is_r ? dgb : (is_g ? dbr : drg);
    const float off = is_r ? 0.0f : (is_g ? 2.0f : 4.0f);
    // chroma == 0 => all channels equal => n == 0 and is_r => hue = 60*(0+0) = 0 as required;
    // the denominator only has to be non-zero there (chroma + 1e-30 == chroma otherwise).
    const float qn = quot_neg(n, chroma + k.tiny);
    // (off + q) * 60 == (-q - off) * -60 (RN is sign-symmetric; off == 0, q == +0 gives -0 * -60 = +0)
    const float hue = (qn - off) * k.neg_k60;
    HsvN o;
    // `if hue < 0 { hue += 360 }`; hue is never -0.0 (equal channels give +0), P4: hue % 360 == hue
    o.h = wrap_up(hue, k);
    o.ns = quot_neg(chroma, value + k.tiny); // value == 0 => chroma == 0 => -0; P4: clamps are identities
    o.v = value;
    return o;
}

__device__ __forceinline__ Hsv from_unit_rgb_fast(float r, float g, float b, const FastConsts &k)
{
    const HsvN n = from_unit_rgb_fast_n(r, g, b, k);
    return Hsv{n.h, -n.ns, n.v};
}

// the same from the byte values as floats (fR = (float)R ...)
__device__ __forceinline__ HsvN from_rgb_fast_n(float fR, float fG, float fB, const FastConsts &k)
{
    return from_unit_rgb_fast_n(div255(fR, k), div255(fG, k), div255(fB, k), k);
}

__device__ __forceinline__ Hsv from_rgb_fast(float fR, float fG, float fB, const FastConsts &k)
{
    return from_unit_rgb_fast(div255(fR, k), div255(fG, k), div255(fB, k), k);
}

// hsvfilter/imp.rs:102-115 for finite settings; NEG_SHIFT selects -360 <= shift < 0 vs 0 <= shift <= 360
template <bool NEG_SHIFT>
__device__ __forceinline__ Hsv filter_hsv_fast(const HsvN in, const FastConsts &k)
{
    Hsv hsv;
    const float x = in.h + k.hue_shift;
    if constexpr (NEG_SHIFT) { // x in [-360,360): fmod is the identity, then `if <0 {+=360}`  (P9, P13)
        hsv.h = wrap_up(x, k);
    } else {                   // x in [0,720): subtract 360 iff x >= 360  (P9, P13)
        hsv.h = wrap_down(x, k);
    }
    // finite settings => no NaN => the VOP3 clamp == max(.,0).min(1)
    hsv.s = add_clamp01(k.neg_saturation_mul * in.ns, k.saturation_off); // (-mul)*(-s) == mul*s, signed zeros included
    hsv.v = add_clamp01(k.value_mul * in.v, k.value_off);
    return hsv;
}

// Sextant table for to_rgb_fast: v_perm_b32 selector that places (R,G,B) of sextant k = floor(h/60)
// out of the candidate dword T = [Vc, Vx, V0, *] (selector values 4,5,6) and keeps the
// alpha / x byte of the source pixel (selector values 0..3).  The reference's ladder
// (hsvutils.rs:138-154) uses closed upper bounds; on the boundaries (h' integer) both neighbours
// give the same triple because x is 0 or c there, so half-open sextants [k,k+1) are equivalent;
// k = 6 only occurs for h == 360 where x == 0 (Vx == V0).
//   k:      0      1      2      3      4      5      6,7
//   R,G,B:  c,x,0  x,c,0  0,c,x  0,x,c  x,0,c  c,0,x  c,x,0
// OFF = index of the first colour byte in the pixel, BGR = byte order of the triple.
__device__ __forceinline__ uint32_t sextant_selector(uint32_t k, int off, bool bgr)
{
    // selector byte of R, G, B for sextant k, packed as 0x00BBGGRR
    const uint32_t rgb = k == 1 ? 0x060405u : k == 2 ? 0x050406u : k == 3 ? 0x040506u
                       : k == 4 ? 0x040605u : k == 5 ? 0x050604u : 0x060504u;
    const uint32_t r = rgb & 0xffu, g = (rgb >> 8) & 0xffu, b = rgb >> 16;
    const uint32_t c0 = bgr ? b : r, c2 = bgr ? r : b;
    return off == 0 ? (c0 | (g << 8) | (c2 << 16) | (3u << 24)) : (0u | (c0 << 8) | (g << 16) | (c2 << 24));
}

// hsvutils.rs:132-163 for h in {0} U [1e-30,360], s,v in [0,1].  Writes T = [Vc, Vx, V0, 0] (the
// three candidate channel bytes) and returns the BYTE offset (4*k, k in 0..7) of the sextant's selector
// in the 8-entry table.
__device__ __forceinline__ uint32_t to_rgb_fast(const Hsv in, const FastConsts &k, uint32_t &T)
{
    const float c = in.v * in.s;
    // hh = RN(h/120) == hp/2 exactly (P11); the sextant comes out of the mantissa of hh + (2^20 - 1/16):
    // ulp is 1/8 there, so bits 2..4 hold floor(2*hh) = floor(hp) (P12; on integer hp the neighbour it may
    // name instead yields the same triple, entries 6,7 of the table repeat entry 0)
    const float hh = fmac_sv(in.h * k.c120lo, k.c120, in.h);
    const float f = __builtin_amdgcn_fractf(hh);          // P3: fmod(hp,2) == 2*fract(hp/2)
    const uint32_t sel_off = __float_as_uint(hh + k.sext_magic) & 28u;
    // 1 - |RN(2f - 1)| == fma(-2, |f - 0.5|, 1): RN(2f-1) == 2*RN(f-0.5) (scaling by 2 is exact)  (P14)
    const float x = c * __builtin_fmaf(-2.0f, fabsf(f - 0.5f), 1.0f);
    const float m = in.v - c;
    // (p + m) * 255 with p in {c, x, 0}; all lie in [0,255], `as u8` truncates
    const float yc = (c + m) * k.k255, yx = (x + m) * k.k255, y0 = m * k.k255;
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(T) : "v"(yc));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(yx));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(T) : "v"(y0));
    return sel_off;
}

__device__ __forceinline__ uint32_t sextant_at(const uint32_t *lut, uint32_t byte_off)
{
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lut) + byte_off);
}

// One pixel of hsvfilter from r,g,b = RN(byte/255): candidate dword T and sextant out.
template <bool NEG_SHIFT>
__device__ __forceinline__ uint32_t hsvfilter_fast_unit(float r, float g, float b, const FastConsts &k, uint32_t &T)
{
    return to_rgb_fast(filter_hsv_fast<NEG_SHIFT>(from_unit_rgb_fast_n(r, g, b, k), k), k, T);
}

// One pixel of hsvfilter: byte values as floats in, candidate dword T and sextant out.
template <bool NEG_SHIFT>
__device__ __forceinline__ uint32_t hsvfilter_fast(float fR, float fG, float fB, const FastConsts &k, uint32_t &T)
{
    return to_rgb_fast(filter_hsv_fast<NEG_SHIFT>(from_rgb_fast_n(fR, fG, fB, k), k), k, T);
}

// hsvdetector/imp.rs:141-155 for finite settings with |ref_hue_offset| <= 360: the `+= 360` and the
// `% 360` become sign-mask +-360 (P9: x in [-360,720) after one conditional +360 lies in [0,720)),
// each `|a - ref| <= var` becomes the sign of RN(var - |a - ref|) (exact sign, +0 on equality; the
// host maps a -0.0 var to +0.0).  Returns 0xffffffff for a MISS and 0 for a hit.
__device__ __forceinline__ uint32_t detect_miss_mask_fast(const HsvN hsv, const HsvDetectorParams &p)
{
    const FastConsts &k = p.consts;
    const float x = hsv.h + p.ref_hue_offset;
    const float x1 = wrap_up(x, k);
    const float x2 = wrap_down(x1, k);
    const float d1 = p.hue_var - fabsf(x2 - p.k180);
    const float d2 = p.saturation_var - fabsf(hsv.ns + p.saturation_ref); // |s - ref| == |-s + ref|
    const float d3 = p.value_var - fabsf(hsv.v - p.value_ref);
    uint32_t m = __float_as_uint(d1) | __float_as_uint(d2) | __float_as_uint(d3);
    asm("v_ashrrev_i32 %0, 31, %0" : "+v"(m));
    return m;
}

__device__ __forceinline__ uint32_t detect_miss_mask_fast(const Hsv hsv, const HsvDetectorParams &p)
{
    return detect_miss_mask_fast(HsvN{hsv.h, -hsv.s, hsv.v}, p);
}

// ---------------------------------------------------------------- dispatch helpers

template <int VARIANT>
__device__ __forceinline__ Hsv from_rgb(uint32_t R, uint32_t G, uint32_t B, const FastConsts &k)
{
    if constexpr (VARIANT == kGeneral)
        return from_rgb_general(R, G, B);
    else
        return from_rgb_fast((float)R, (float)G, (float)B, k);
}

// (R,G,B) -> (R,G,B) of hsvfilter for one pixel.  VARIANT: kGeneral, kFast (0 <= shift <= 360),
// kFastNeg (-360 <= shift < 0).  `lut` = 8 sextant selectors in LDS for layout (off 0, RGB).
template <int VARIANT>
__device__ __forceinline__ void hsvfilter_pixel(uint32_t &R, uint32_t &G, uint32_t &B, const FastConsts &k,
                                                const uint32_t *lut)
{
    if constexpr (VARIANT == kGeneral) {
        const HsvFilterParams p{k.hue_shift, k.saturation_mul, k.saturation_off, k.value_mul, k.value_off};
        to_rgb_general(filter_hsv_general(from_rgb_general(R, G, B), p), R, G, B);
    } else {
        uint32_t T;
        const uint32_t sel_off = hsvfilter_fast<VARIANT == kFastNeg>((float)R, (float)G, (float)B, k, T);
        const uint32_t rot = __builtin_amdgcn_perm(T, 0u, sextant_at(lut, sel_off));
        R = rot & 0xffu;
        G = (rot >> 8) & 0xffu;
        B = (rot >> 16) & 0xffu;
    }
}

} // namespace mvfx
