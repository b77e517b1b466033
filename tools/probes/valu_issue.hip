// How many shader cycles one wave64 VALU instruction occupies its SIMD for on gfx950, per instruction class.
// Every kernel is the same loop around 64 inline-asm instructions on 8 independent accumulators; 8 waves per SIMD hide the
// dependent-issue latency, so cycles / (waves per SIMD x instructions per wave) is the issue cost.  Cycles from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/valu_issue.hip -o gpurun_out/valu_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(name, ASM)                                                                              \
    __global__ __launch_bounds__(256) void name(uint64_t *out, float *sink, int iters, float fa, float fb) { \
        float a[8];                                                                                    \
        for (int i = 0; i < 8; i++) a[i] = fa + threadIdx.x * 0.001f + i;                              \
        float b = fb, c = fa * 0.5f;                                                                   \
        uint64_t t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; it++) {                                                           \
            BODY64(ASM)                                                                                \
        }                                                                                              \
        uint64_t t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                   \
        for (int i = 0; i < 8; i++) s += a[i];                                                         \
        if (s == 12345.678f) sink[0] = s;                                                              \
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1; }                                               \
    }

#define I_FMA(i)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_ADD(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_MUL(i)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_MAX(i)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_MAX3(i)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_ADDU(i)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define I_PERM(i)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
#define I_CVTU(i)  asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
#define I_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define I_RCP(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define I_CNDM(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define I_CMP(i)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define I_CMPS(i)  asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
#define I_SDWA(i)  asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[i]) : "v"(pb));
#define I_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_SAD(i)   asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_FMAK(i)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_MED3(i)  asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_FMAS(i)  asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(a[i]) : "v"(c));
#define I_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(db));

KERNEL(k_fma, I_FMA) KERNEL(k_add, I_ADD) KERNEL(k_mul, I_MUL) KERNEL(k_max, I_MAX) KERNEL(k_max3, I_MAX3) KERNEL(k_addu, I_ADDU)
KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_perm, I_PERM) KERNEL(k_cvtub, I_CVTUB) KERNEL(k_cvtu, I_CVTU) KERNEL(k_floor, I_FLOOR)
KERNEL(k_rcp, I_RCP) KERNEL(k_cndm, I_CNDM) KERNEL(k_cmp, I_CMP) KERNEL(k_cmps, I_CMPS) KERNEL(k_sdwa, I_SDWA) KERNEL(k_mov, I_MOV)
KERNEL(k_mullo, I_MULLO) KERNEL(k_madu24, I_MADU24) KERNEL(k_sad, I_SAD) KERNEL(k_fmac, I_FMAK) KERNEL(k_med3, I_MED3) KERNEL(k_fmas, I_FMAS)

#define I_k_subf(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_subf, I_k_subf)
#define I_k_minf(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_minf, I_k_minf)
#define I_k_and(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_and, I_k_and)
#define I_k_or(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_or, I_k_or)
#define I_k_xor(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_xor, I_k_xor)
#define I_k_shl(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_shl, I_k_shl)
#define I_k_shr(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_shr, I_k_shr)
#define I_k_subu(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_subu, I_k_subu)
#define I_k_bfe(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_bfe, I_k_bfe)
#define I_k_andor(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_andor, I_k_andor)
#define I_k_add3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_add3, I_k_add3)
#define I_k_cvtpk(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cvtpk, I_k_cvtpk)
#define I_k_cvtfu(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cvtfu, I_k_cvtfu)
#define I_k_cvtub0(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cvtub0, I_k_cvtub0)
#define I_k_fract(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fract, I_k_fract)
#define I_k_rndne(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_rndne, I_k_rndne)
#define I_k_addlit(i) asm volatile("v_add_f32 %0, 0x40490fdb, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_addlit, I_k_addlit)
#define I_k_addinl(i) asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_addinl, I_k_addinl)
#define I_k_mulinl(i) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mulinl, I_k_mulinl)
#define I_k_mullit(i) asm volatile("v_mul_f32 %0, 0x40490fdb, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mullit, I_k_mullit)
#define I_k_addsg(i) asm volatile("v_add_f32 %0, s20, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_addsg, I_k_addsg)
#define I_k_fmaak(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x40490fdb" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fmaak, I_k_fmaak)
#define I_k_fmamk(i) asm volatile("v_fmamk_f32 %0, %0, 0x40490fdb, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fmamk, I_k_fmamk)
#define I_k_fmainl(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fmainl, I_k_fmainl)
#define I_k_fmaneg(i) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fmaneg, I_k_fmaneg)
#define I_k_mulabs(i) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mulabs, I_k_mulabs)
#define I_k_addclamp(i) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_addclamp, I_k_addclamp)
#define I_k_mulomod(i) asm volatile("v_mul_f32_e64 %0, %0, %1 mul:2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mulomod, I_k_mulomod)
#define I_k_maxu(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_maxu, I_k_maxu)
#define I_k_minu(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_minu, I_k_minu)
#define I_k_mulu24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mulu24, I_k_mulu24)
#define I_k_cndms(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cndms, I_k_cndms)
#define I_k_cndmd(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cndmd, I_k_cndmd)
#define I_k_dpp(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_dpp, I_k_dpp)
#define I_k_adddpp(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_adddpp, I_k_adddpp)
#define I_k_addco(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
KERNEL(k_addco, I_k_addco)
#define I_k_cvth(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cvth, I_k_cvth)
#define I_k_pkaddh(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkaddh, I_k_pkaddh)
#define I_k_pkfmah(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkfmah, I_k_pkfmah)
#define I_k_pkmaxh(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkmaxh, I_k_pkmaxh)
#define I_k_pkaddu(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkaddu, I_k_pkaddu)
#define I_k_pkmulu(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkmulu, I_k_pkmulu)
#define I_k_pkmadu(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkmadu, I_k_pkmadu)
#define I_k_dot4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_dot4, I_k_dot4)
#define I_k_ldexp(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_ldexp, I_k_ldexp)
#define I_k_mixfp(i) if ((i) & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixfp, I_k_mixfp)
#define I_k_mixam(i) if ((i) & 1) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixam, I_k_mixam)
#define I_k_mixfc(i) if ((i) & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixfc, I_k_mixfc)
#define I_k_mixfr(i) if ((i) & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixfr, I_k_mixfr)
#define I_k_mixf3r(i) if (((i) & 3) == 3) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixf3r, I_k_mixf3r)


#define I_k_cndm_e64vcc(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cndm_e64vcc, I_k_cndm_e64vcc)
#define I_k_cmpcnd(i) if ((i) == 0) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); } else { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cmpcnd, I_k_cmpcnd)
#define I_k_cmpcnd2(i) if (((i) & 1) == 0) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); } else { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cmpcnd2, I_k_cmpcnd2)
#define I_k_cmpcnd2s(i) if (((i) & 1) == 0) { asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21"); } else { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cmpcnd2s, I_k_cmpcnd2s)
#define I_k_mulsg(i) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_mulsg, I_k_mulsg)
#define I_k_fmacsg(i) asm volatile("v_fmac_f32 %0, s20, %1" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_fmacsg, I_k_fmacsg)
#define I_k_cvtsdwa(i) asm volatile("v_cvt_u32_f32_sdwa %0, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_cvtsdwa, I_k_cvtsdwa)
#define I_k_min3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_min3, I_k_min3)
#define I_k_or3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_or3, I_k_or3)
#define I_k_mixpc(i) if ((i) & 1) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixpc, I_k_mixpc)
#define I_k_mixpr(i) if ((i) & 1) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixpr, I_k_mixpr)
#define I_k_mix3f1p(i) if (((i) & 3) == 3) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mix3f1p, I_k_mix3f1p)
#define I_k_mix1f3p(i) if (((i) & 3) != 3) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mix1f3p, I_k_mix1f3p)
#define I_k_mixfsg(i) if ((i) & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_mul_f32 %0, s20, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_mixfsg, I_k_mixfsg)
#define I_k_salu(i) asm volatile("s_add_u32 s20, s20, s21" : : : "s20", "scc");
KERNEL(k_salu, I_k_salu)
#define I_k_mixfsalu(i) if ((i) & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("s_add_u32 s20, s20, s21" : : : "s20", "scc"); }
KERNEL(k_mixfsalu, I_k_mixfsalu)
#define I_k_mixpsalu(i) if ((i) & 1) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("s_add_u32 s20, s20, s21" : : : "s20", "scc"); }
KERNEL(k_mixpsalu, I_k_mixpsalu)


#define I_k_cndfma(i) if ((i) & 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cndfma, I_k_cndfma)
#define I_k_cmpcndfma(i) if ((i) == 0) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); } else if ((i) & 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cmpcndfma, I_k_cmpcndfma)
#define I_k_cndperm(i) if ((i) & 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cndperm, I_k_cndperm)
#define I_k_cnd3fma(i) if (((i) & 3) == 3) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cnd3fma, I_k_cnd3fma)
#define I_k_cnde64e32(i) if ((i) & 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c)); } else { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL(k_cnde64e32, I_k_cnde64e32)
#define I_k_addcchain(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
KERNEL(k_addcchain, I_k_addcchain)

typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL2(name, ASM)                                                                             \
    __global__ __launch_bounds__(256) void name(uint64_t *out, float *sink, int iters, float fa, float fb) { \
        float a[8]; f2 pa[8];                                                                          \
        for (int i = 0; i < 8; i++) { a[i] = fa + threadIdx.x * 0.001f + i; pa[i] = f2{a[i], fa}; }    \
        float b = fb, c = fa * 0.5f; f2 pb = {fb, fb};                                                 \
        uint64_t t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; it++) {                                                           \
            BODY64(ASM)                                                                                \
        }                                                                                              \
        uint64_t t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                   \
        for (int i = 0; i < 8; i++) s += a[i] + pa[i].x + pa[i].y;                                     \
        if (s == 12345.678f) sink[0] = s;                                                              \
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1; }              \
    }
#define I_k_pkmul(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb));
KERNEL2(k_pkmul, I_k_pkmul)
#define I_k_pkaddclamp(i) asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(pa[i]) : "v"(pb));
KERNEL2(k_pkaddclamp, I_k_pkaddclamp)
#define I_k_pkmulsel(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(pa[i]) : "v"(pb));
KERNEL2(k_pkmulsel, I_k_pkmulsel)
#define I_k_mixpkf(i) if ((i) & 1) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb)); } else { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
KERNEL2(k_mixpkf, I_k_mixpkf)
#define I_k_mixpk2f(i) if (((i) % 3) == 0) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb)); } else { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
KERNEL2(k_mixpk2f, I_k_mixpk2f)
#define I_k_mixpkfma(i) if ((i) & 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[i]) : "v"(pb)); } else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
KERNEL2(k_mixpkfma, I_k_mixpkfma)
#define I_k_mixpkcvt(i) if ((i) & 1) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb)); } else { asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i])); }
KERNEL2(k_mixpkcvt, I_k_mixpkcvt)

__global__ __launch_bounds__(256) void k_pkfma(uint64_t *out, float *sink, int iters, float fa, float fb) {
    f2 pa[8];
    for (int i = 0; i < 8; i++) pa[i] = f2{fa + threadIdx.x * 0.001f + i, fa};
    f2 pb = {fb, fb};
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) { BODY64(I_PKFMA) }
    uint64_t t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += pa[i].x + pa[i].y;
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1; }
}
__global__ __launch_bounds__(256) void k_add64(uint64_t *out, float *sink, int iters, float fa, float fb) {
    double da[8];
    for (int i = 0; i < 8; i++) da[i] = fa + threadIdx.x * 0.001 + i;
    double db = fb;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) { BODY64(I_ADD64) }
    uint64_t t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; i++) s += da[i];
    if (s == 12345.678) sink[0] = (float)s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1; }
}

typedef void (*kern_t)(uint64_t *, float *, int, float, float);

int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 (sgpr src)", k_fmas}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul},
        {"v_max_f32", k_max}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_add_u32", k_addu}, {"v_lshl_or_b32", k_lshlor},
        {"v_perm_b32", k_perm}, {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_u32_f32", k_cvtu}, {"v_floor_f32", k_floor}, {"v_rcp_f32", k_rcp},
        {"v_cndmask_b32 (vcc)", k_cndm}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr pair", k_cmps}, {"v_add_u32_sdwa", k_sdwa},
        {"v_mov_b32", k_mov}, {"v_mul_lo_u32", k_mullo}, {"v_mad_u32_u24", k_madu24}, {"v_sad_u8", k_sad}, {"v_pk_fma_f32", k_pkfma},
        {"v_add_f64", k_add64},
        {"v_sub_f32", k_subf},
        {"v_min_f32", k_minf},
        {"v_and_b32", k_and},
        {"v_or_b32", k_or},
        {"v_xor_b32", k_xor},
        {"v_lshlrev_b32 (imm)", k_shl},
        {"v_lshrrev_b32 (imm)", k_shr},
        {"v_sub_u32", k_subu},
        {"v_bfe_u32", k_bfe},
        {"v_and_or_b32", k_andor},
        {"v_add3_u32", k_add3},
        {"v_cvt_pk_u8_f32", k_cvtpk},
        {"v_cvt_f32_u32", k_cvtfu},
        {"v_cvt_f32_ubyte0", k_cvtub0},
        {"v_fract_f32", k_fract},
        {"v_rndne_f32", k_rndne},
        {"v_add_f32 (literal)", k_addlit},
        {"v_add_f32 (inline 0.5)", k_addinl},
        {"v_mul_f32 (inline 2.0)", k_mulinl},
        {"v_mul_f32 (literal)", k_mullit},
        {"v_add_f32 (sgpr)", k_addsg},
        {"v_fmaak_f32 (literal)", k_fmaak},
        {"v_fmamk_f32 (literal)", k_fmamk},
        {"v_fma_f32 (inline 1.0)", k_fmainl},
        {"v_fma_f32 (neg mod)", k_fmaneg},
        {"v_mul_f32_e64 (abs mod)", k_mulabs},
        {"v_add_f32_e64 (clamp)", k_addclamp},
        {"v_mul_f32_e64 (mul:2)", k_mulomod},
        {"v_max_u32", k_maxu},
        {"v_min_u32", k_minu},
        {"v_mul_u32_u24", k_mulu24},
        {"v_cndmask_b32 e64 sgpr", k_cndms},
        {"v_cndmask_b32 dst!=src", k_cndmd},
        {"v_mov_b32 dpp row_shr:1", k_dpp},
        {"v_add_f32 dpp", k_adddpp},
        {"v_add_co_u32 vcc", k_addco},
        {"v_cvt_f16_f32", k_cvth},
        {"v_pk_add_f16", k_pkaddh},
        {"v_pk_fma_f16", k_pkfmah},
        {"v_pk_max_f16", k_pkmaxh},
        {"v_pk_add_u16", k_pkaddu},
        {"v_pk_mul_lo_u16", k_pkmulu},
        {"v_pk_mad_u16", k_pkmadu},
        {"v_dot4_u32_u8", k_dot4},
        {"v_ldexp_f32", k_ldexp},
        {"mix fma,perm alternating", k_mixfp},
        {"mix add_f32,max_f32 altern.", k_mixam},
        {"mix fma,cvt_f32_ubyte altern.", k_mixfc},
        {"mix fma,rcp alternating", k_mixfr},
        {"mix fma x3 + rcp x1", k_mixf3r},
        {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32 clamp", k_pkaddclamp}, {"v_pk_mul_f32 op_sel hi,hi", k_pkmulsel},
        {"mix pk_mul, add_f32 altern.", k_mixpkf}, {"mix pk_mul x1 + add_f32 x2", k_mixpk2f}, {"mix pk_fma, fma altern.", k_mixpkfma},
        {"mix pk_mul, cvt alternating", k_mixpkcvt},

        {"mix cndmask e32, fma altern.", k_cndfma}, {"cmp + (cndmask e32, fma) x", k_cmpcndfma}, {"mix cndmask e32, perm alt.", k_cndperm},
        {"mix fma x3 + cndmask e32 x1", k_cnd3fma}, {"mix cndmask e64 sgpr, e32 vcc", k_cnde64e32}, {"v_addc_co_u32 vcc chain", k_addcchain},

        {"v_cndmask_b32_e64 vcc", k_cndm_e64vcc}, {"v_cmp vcc + 7 cndmask e32", k_cmpcnd}, {"v_cmp vcc, cndmask e32 alt.", k_cmpcnd2},
        {"v_cmp sgpr, cndmask e64 alt.", k_cmpcnd2s}, {"v_mul_f32 (sgpr)", k_mulsg}, {"v_fmac_f32 (sgpr)", k_fmacsg},
        {"v_cvt_u32_f32_sdwa", k_cvtsdwa}, {"v_min3_f32", k_min3}, {"v_or3_b32", k_or3}, {"mix perm,cvt alternating", k_mixpc},
        {"mix perm,rcp alternating", k_mixpr}, {"mix fma x3 + perm x1", k_mix3f1p}, {"mix fma x1 + perm x3", k_mix1f3p},
        {"mix fma, mul(sgpr) altern.", k_mixfsg}, {"s_add_u32", k_salu}, {"mix fma, s_add_u32 altern.", k_mixfsalu}, {"mix perm, s_add_u32 alt.", k_mixpsalu},

    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 512;
    uint64_t *out;
    float *sink;
    hipMalloc(&out, sizeof(uint64_t) * cus * 16);
    hipMalloc(&sink, 64);
    printf("# %s, %d CUs, clockRate %d kHz; 64 x %d instructions per wave\n", prop.gcnArchName, cus, prop.clockRate, iters);
    printf("%-30s %s\n", "instruction",
           "wall ns per wave64 instruction per SIMD at 1, 2, 4, 8 waves per SIMD | s_memtime ticks per ns | ticks per instruction of one wave at 1 and 8 waves per SIMD");
    for (auto &e : ks) {
        printf("%-30s", e.name);
        double rate = 0, per1 = 0, per8 = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            e.k<<<blocks, 256>>>(out, sink, iters, 1.0f, 1.0001f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            e.k<<<blocks, 256>>>(out, sink, iters, 1.0f, 1.0001f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(2 * blocks);
            hipMemcpy(h.data(), out, sizeof(uint64_t) * 2 * blocks, hipMemcpyDeviceToHost);
            uint64_t lo = ~0ull, hi = 0;
            std::vector<uint64_t> d(blocks);
            for (int b = 0; b < blocks; b++) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); d[b] = h[2 * b + 1] - h[2 * b]; }
            std::sort(d.begin(), d.end());
            const double insts = 64.0 * iters;
            printf("  %6.3f", ms * 1e6 / (insts * wps));
            rate = (hi - lo) / (ms * 1e6);
            if (wps == 1) per1 = d[blocks / 2] / insts;
            if (wps == 8) per8 = d[blocks / 2] / insts;
        }
        printf("  | %5.3f | %6.2f %6.2f\n", rate, per1, per8);
    }
    return 0;
}
