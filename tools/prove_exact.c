/*
 * tools/prove_exact.c -- exhaustive CPU proofs for the exact strength reductions used by
 * the FAST hsv kernels (gst-plugin-rs_amd/csrc/hsv_math.hpp).  IEEE-754 binary32 mul/fma are
 * deterministic, so a proof on the host FPU (compiled with -ffp-contract=off, fmaf() mapped to
 * a hardware FMA by -mfma) carries over to v_mul_f32 / v_fma_f32 on gfx950.
 *
 *   gcc -O2 -ffp-contract=off -mfma tools/prove_exact.c -o /tmp/prove_exact -lm && /tmp/prove_exact
 *
 * Claims checked (each prints PASS/FAIL):
 *  P1  u8/255.0f      == fma(fma(-255,q0,x), C255, q0), q0 = x*C255, for x = 0..255
 *  P1b u8/255.0f      == x * C for a single constant C (cheaper form), if one exists
 *  P2  h/60.0f        == fma(fma(-60,q0,h), C60, q0),   q0 = h*C60, for h = 0 and every float h
 *      in [1e-30,360]  (below ~4.7e-38 the quotient is denormal and the residual underflows;
 *      the FAST kernel's host-side gate keeps h out of (0,1e-30), see hsv_math.hpp)
 *  P3  fmodf(hp,2)    == 2*(t - floorf(t)), t = 0.5*hp, for hp = 0 and every float hp in [1e-32,6]
 *  P8  2-op forms: u8/255 == fma(x, C, x*Clo) and h/60 == fma(h, C60, h*C60lo) and u16/65535 likewise,
 *      with Clo = (float)(1/255 - (double)C) etc.  (same domains as P1, P2, P6)
 *  P9  hue wrap by sign masks: for x in [0,720): x >= 360  <=>  signbit(PRED360 - x), PRED360 = 359.99997
 *      (largest float below 360), and the result x - (mask & 360) equals P7's; for x in [-360,360):
 *      signbit(x) ? x + 360 : x equals the reference (x = -0.0 does not occur, see hsv_math.hpp)
 *  P11 h/120: fma(h, C60/2, h*(C60lo/2)) == 0.5f*(h/60.0f) bit for bit (same domain as P2)
 *  P12 sextant from the mantissa: (bits(hh + (2^20 - 1/16)) & 28) / 4, hh = h/120, names the same v_perm
 *      selector as floor(h/60) (entries 6,7 == entry 0; on integer h/60 either neighbour is valid)
 *  P13 hue wraps as unsigned-integer minima: x in [0,720): fmod(x,360) == min_u32(bits(x-360), bits(x));
 *      x in [-360,360) \ {-0}: (x<0 ? x+360 : x) == min_u32(bits(x), bits(x+360))
 *  P14 1 - |fma(f,2,-1)| == fma(-2, |f-0.5|, 1) for every float f in [0,1)
 *  P10 f32::round (half away from zero) of v in [0,65535.5] == truncf(v + PRED_HALF), PRED_HALF = 0.49999997
 *      (largest float below 0.5); used by colorlut's float_to_u8 / float_to_u16
 *  P15 float_to_u8 with ONE fused operation: for every float v in [0,1], roundf(v * 255.0f) == truncf(fmaf(v, 255.0f, PRED_HALF))
 *      (colorlut_xtile_kernel; the same with 65535 fails on 2 floats -- there fmaf(v, 65535.0f, 0.5f) is the exact one)
 *  P7  fmodf(x,360) followed by `if <0 {+=360}` == conditional +-360 for every float x in
 *      [-360,720) (results compared as floats, +0 == -0)
 *  P4  from_rgb hue is in [0,360) for all 2^24 (R,G,B) => `hue % 360` is the identity
 *      and saturation/value are already inside [0,1] => the clamps are identities
 *  P5  |value - ch| < 1e-5  <=>  CH == max(R,G,B)   for all byte pairs
 *  P6  u16/65535.0f   == fma(fma(-65535,q0,x), C65535, q0) for x = 0..65535
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void)
{
    int ok_all = 1;
    const float C255 = 1.0f / 255.0f, C60 = 1.0f / 60.0f, C65535 = 1.0f / 65535.0f;

    { /* P1 */
        int bad = 0;
        for (int x = 0; x < 256; x++) {
            float fx = (float)x, q0 = fx * C255;
            float q = fmaf(fmaf(-255.0f, q0, fx), C255, q0);
            if (f2u(q) != f2u(fx / 255.0f)) bad++;
        }
        printf("P1  div255 mul+fma+fma: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        ok_all &= !bad;
    }
    { /* P1b */
        uint32_t base = f2u(C255);
        int found = 0;
        for (int d = -2; d <= 2; d++) {
            float c = u2f(base + d);
            int bad = 0;
            for (int x = 0; x < 256; x++)
                if (f2u((float)x * c) != f2u((float)x / 255.0f)) bad++;
            printf("P1b single-mul constant 0x%08x: %d mismatches\n", base + d, bad);
            if (!bad) found = 1;
        }
        printf("P1b single-constant form exists: %s\n", found ? "YES" : "NO");
    }
    { /* P2 */
        uint64_t bad = 0, n = 0;
        for (uint32_t u = f2u(1e-30f) - 1; u <= f2u(360.0f); u++) {
            float h = (u == f2u(1e-30f) - 1) ? 0.0f : u2f(u), q0 = h * C60;
            float q = fmaf(fmaf(-60.0f, q0, h), C60, q0);
            if (f2u(q) != f2u(h / 60.0f)) bad++;
            n++;
        }
        printf("P2  div60 over %llu floats in {0} U [1e-30,360]: %s (%llu mismatches)\n",
               (unsigned long long)n, bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P3 */
        uint64_t bad = 0, n = 0;
        for (uint32_t u = f2u(1e-32f) - 1; u <= f2u(6.0f); u++) {
            float hp = (u == f2u(1e-32f) - 1) ? 0.0f : u2f(u), t = 0.5f * hp;
            float m = 2.0f * (t - floorf(t));
            if (f2u(m) != f2u(fmodf(hp, 2.0f))) bad++;
            n++;
        }
        printf("P3  fmod(hp,2) via fract over %llu floats in {0} U [1e-32,6]: %s (%llu mismatches)\n",
               (unsigned long long)n, bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P4 + P5 */
        uint64_t bad_h = 0, bad_sv = 0, bad_eps = 0;
        float hmax = 0.0f;
        for (uint32_t i = 0; i < (1u << 24); i++) {
            uint32_t R = i & 255, G = (i >> 8) & 255, B = i >> 16;
            uint32_t mx = R > G ? (R > B ? R : B) : (G > B ? G : B);
            uint32_t mn = R < G ? (R < B ? R : B) : (G < B ? G : B);
            float r = (float)R / 255.0f, g = (float)G / 255.0f, b = (float)B / 255.0f;
            float value = (float)mx / 255.0f, chroma = value - (float)mn / 255.0f;
            int er = fabsf(value - r) < 0.00001f, eg = fabsf(value - g) < 0.00001f,
                eb = fabsf(value - b) < 0.00001f;
            if (er != (R == mx) || eg != (G == mx) || eb != (B == mx)) bad_eps++;
            float hue;
            if (chroma == 0.0f) hue = 0.0f;
            else if (er) hue = 60.0f * ((g - b) / chroma);
            else if (eg) hue = 60.0f * (2.0f + ((b - r) / chroma));
            else if (eb) hue = 60.0f * (4.0f + ((r - g) / chroma));
            else hue = 0.0f;
            if (hue < 0.0f) hue += 360.0f;
            if (!(hue >= 0.0f && hue < 360.0f)) bad_h++;
            if (hue > hmax) hmax = hue;
            float sat = value == 0.0f ? 0.0f : chroma / value;
            if (!(sat >= 0.0f && sat <= 1.0f && value >= 0.0f && value <= 1.0f)) bad_sv++;
        }
        printf("P4  hue in [0,360) (max %.9g): %s; sat,value in [0,1]: %s\n", hmax,
               bad_h ? "FAIL" : "PASS", bad_sv ? "FAIL" : "PASS");
        printf("P5  epsilon test == integer max test: %s\n", bad_eps ? "FAIL" : "PASS");
        ok_all &= !bad_h && !bad_sv && !bad_eps;
    }
    { /* P6 */
        int bad = 0;
        for (int x = 0; x < 65536; x++) {
            float fx = (float)x, q0 = fx * C65535;
            float q = fmaf(fmaf(-65535.0f, q0, fx), C65535, q0);
            if (f2u(q) != f2u(fx / 65535.0f)) bad++;
        }
        printf("P6  div65535 mul+fma+fma: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        ok_all &= !bad;
    }
    { /* P7 */
        uint64_t bad = 0, n = 0;
        for (int neg = 0; neg < 2; neg++) {
            uint32_t top = neg ? f2u(360.0f) : f2u(720.0f) - 1;
            for (uint32_t u = 0; u <= top; u++) {
                float x = neg ? -u2f(u) : u2f(u);
                float ref = fmodf(x, 360.0f);
                if (ref < 0.0f) ref += 360.0f;
                float t = x >= 360.0f ? x - 360.0f : x;
                float fast = t < 0.0f ? t + 360.0f : t;
                if (!(ref == fast)) bad++;
                n++;
            }
        }
        printf("P7  hue wrap over %llu floats in [-360,720): %s (%llu mismatches)\n",
               (unsigned long long)n, bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P8 */
        const float C255lo = (float)(1.0 / 255.0 - (double)C255), C60lo = (float)(1.0 / 60.0 - (double)C60),
                    C65535lo = (float)(1.0 / 65535.0 - (double)C65535);
        int bad = 0;
        for (int x = 0; x < 256; x++) {
            float fx = (float)x;
            if (f2u(fmaf(fx, C255, fx * C255lo)) != f2u(fx / 255.0f)) bad++;
        }
        for (int x = 0; x < 65536; x++) {
            float fx = (float)x;
            if (f2u(fmaf(fx, C65535, fx * C65535lo)) != f2u(fx / 65535.0f)) bad++;
        }
        uint64_t bad60 = 0;
        for (uint32_t u = f2u(1e-30f) - 1; u <= f2u(360.0f); u++) {
            float h = (u == f2u(1e-30f) - 1) ? 0.0f : u2f(u);
            if (f2u(fmaf(h, C60, h * C60lo)) != f2u(h / 60.0f)) bad60++;
        }
        printf("P8  2-op div255/div65535: %s (%d); 2-op div60: %s (%llu)  [C255lo=%a C60lo=%a C65535lo=%a]\n",
               bad ? "FAIL" : "PASS", bad, bad60 ? "FAIL" : "PASS", (unsigned long long)bad60, C255lo, C60lo, C65535lo);
        ok_all &= !bad && !bad60;
    }
    { /* P9 */
        const float PRED360 = u2f(f2u(360.0f) - 1);
        uint64_t bad = 0;
        for (uint32_t u = 0; u <= f2u(720.0f) - 1; u++) {
            float x = u2f(u);
            float ref = fmodf(x, 360.0f);
            if (ref < 0.0f) ref += 360.0f;
            float w = PRED360 - x;
            uint32_t mask = (uint32_t)((int32_t)f2u(w) >> 31);
            float fast = x - u2f(mask & f2u(360.0f));
            if (!(ref == fast)) bad++;
        }
        for (uint32_t u = 1; u <= f2u(360.0f); u++) { /* negative x, -0.0 excluded */
            float x = -u2f(u);
            float ref = fmodf(x, 360.0f);
            if (ref < 0.0f) ref += 360.0f;
            uint32_t mask = (uint32_t)((int32_t)f2u(x) >> 31);
            float fast = x + u2f(mask & f2u(360.0f));
            if (!(ref == fast)) bad++;
        }
        printf("P9  sign-mask hue wraps: %s (%llu mismatches)\n", bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P10 */
        const float PRED_HALF = u2f(f2u(0.5f) - 1);
        uint64_t bad = 0, n = 0;
        for (uint32_t u = 0; u <= f2u(65535.5f); u++) {
            float v = u2f(u);
            if (truncf(v + PRED_HALF) != roundf(v)) bad++;
            n++;
        }
        printf("P10 round-half-away == trunc(v + pred(0.5)) over %llu floats in [0,65535.5]: %s (%llu)\n",
               (unsigned long long)n, bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P11 + P12 */
        const float C60lo = (float)(1.0 / 60.0 - (double)C60);
        const float C120 = 0.5f * C60, C120lo = 0.5f * C60lo, K = 1048575.9375f;
        uint64_t bad11 = 0, bad12 = 0;
        for (uint32_t u = f2u(1e-30f) - 1; u <= f2u(360.0f); u++) {
            float h = (u == f2u(1e-30f) - 1) ? 0.0f : u2f(u);
            float hp = h / 60.0f, hh = fmaf(h, C120, h * C120lo);
            if (f2u(hh) != f2u(0.5f * hp)) bad11++;
            uint32_t idx = (f2u(hh + K) & 28u) >> 2, k0 = (uint32_t)hp;
            uint32_t ci = idx >= 6 ? 0 : idx, c0 = k0 >= 6 ? 0 : k0, cm = (k0 == 0 ? 0 : (k0 - 1 >= 6 ? 0 : k0 - 1));
            int integer = (hp == (float)k0);
            if (!(ci == c0 || (integer && ci == cm))) bad12++;
        }
        printf("P11 2-op h/120 == 0.5*(h/60): %s (%llu);  P12 mantissa sextant: %s (%llu)\n", bad11 ? "FAIL" : "PASS",
               (unsigned long long)bad11, bad12 ? "FAIL" : "PASS", (unsigned long long)bad12);
        ok_all &= !bad11 && !bad12;
    }
    { /* P13 */
        uint64_t bad = 0;
        for (uint32_t u = 0; u <= f2u(720.0f) - 1; u++) { /* x in [0,720) */
            float x = u2f(u), ref = fmodf(x, 360.0f), y = x - 360.0f;
            uint32_t m = f2u(y) < f2u(x) ? f2u(y) : f2u(x);
            if (f2u(ref) != m) bad++;
            if (u <= f2u(360.0f) - 1) { /* x in [0,360): `if x < 0 { x += 360 }` leaves x */
                float z = x + 360.0f;
                uint32_t m2 = f2u(x) < f2u(z) ? f2u(x) : f2u(z);
                if (m2 != f2u(x)) bad++;
            }
        }
        for (uint32_t u = 1; u <= f2u(360.0f); u++) { /* x in [-360,0), -0.0 excluded */
            float x = -u2f(u), ref = x + 360.0f;
            uint32_t m = f2u(x) < f2u(ref) ? f2u(x) : f2u(ref);
            if (m != f2u(ref)) bad++;
        }
        printf("P13 min_u32 hue wraps: %s (%llu mismatches)\n", bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P14 */
        uint64_t bad = 0;
        for (uint32_t u = 0; u < f2u(1.0f); u++) {
            float f = u2f(u);
            float a = fmaf(f, 2.0f, -1.0f), w0 = 1.0f - fabsf(a);
            float w1 = fmaf(-2.0f, fabsf(f - 0.5f), 1.0f);
            if (f2u(w0) != f2u(w1)) bad++;
        }
        printf("P14 1-|2f-1| == fma(-2,|f-0.5|,1) on [0,1): %s (%llu mismatches)\n", bad ? "FAIL" : "PASS", (unsigned long long)bad);
        ok_all &= !bad;
    }
    { /* P15 */
        const float h = nextafterf(0.5f, 0.0f);
        uint64_t bad8 = 0, bad16 = 0;
        for (uint32_t u = 0; u <= f2u(1.0f); u++) {
            const float v = u2f(u);
            if (truncf(fmaf(v, 255.0f, h)) != roundf(v * 255.0f)) bad8++;
            if (truncf(fmaf(v, 65535.0f, 0.5f)) != roundf(v * 65535.0f)) bad16++;
        }
        printf("P15 fused float_to_u8 trunc(fma(v,255,pred(0.5))): %s (%llu); float_to_u16 trunc(fma(v,65535,0.5)): %s (%llu)\n",
               bad8 ? "FAIL" : "PASS", (unsigned long long)bad8, bad16 ? "FAIL" : "PASS", (unsigned long long)bad16);
        ok_all &= !bad8 && !bad16;
    }
    printf("%s\n", ok_all ? "ALL PASS" : "SOME FAILED");
    return ok_all ? 0 : 1;
}
