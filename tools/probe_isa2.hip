// tools/probe_isa2.hip -- per-instruction issue-rate census on gfx950 (which VALU ops are in the
// fast fp32 class, which are 4-cycle).  8 independent registers per lane, 8 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_isa2.hip -o /tmp/probe_isa2 && /tmp/probe_isa2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

#define DEFINE_KERNEL(NAME, LINE)                                                              \
    __global__ void NAME(float *out, float a, float b, int iters)                             \
    {                                                                                          \
        float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 2 + threadIdx.x,            \
              x3 = b * 3 + threadIdx.x, x4 = x0 + 5, x5 = x1 + 6, x6 = x2 + 7, x7 = x3 + 8;   \
        unsigned long long m = __ballot(threadIdx.x & 1);                                      \
        for (int i = 0; i < iters; i++) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; u++)                                      \
            {                                                                                  \
                asm volatile(LINE                                                              \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5),     \
                               "+v"(x6), "+v"(x7)                                              \
                             : "v"(a), "v"(b), "s"(m)                                          \
                             : "vcc");                                                         \
            }                                                                                  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;    \
    }

#define L1(op, i) op " %" #i ", %" #i "\n"
#define L2(op, i) op " %" #i ", %" #i ", %8\n"
#define L3(op, i) op " %" #i ", %" #i ", %8, %9\n"

#define K_OP1(NAME, op) DEFINE_KERNEL(NAME, L1(op,0) L1(op,1) L1(op,2) L1(op,3) L1(op,4) L1(op,5) L1(op,6) L1(op,7))
#define K_OP2(NAME, op) DEFINE_KERNEL(NAME, L2(op,0) L2(op,1) L2(op,2) L2(op,3) L2(op,4) L2(op,5) L2(op,6) L2(op,7))
#define K_OP3(NAME, op) DEFINE_KERNEL(NAME, L3(op,0) L3(op,1) L3(op,2) L3(op,3) L3(op,4) L3(op,5) L3(op,6) L3(op,7))

K_OP3(k_fma, "v_fma_f32")
K_OP2(k_fmac, "v_fmac_f32")
K_OP2(k_mul, "v_mul_f32")
K_OP2(k_add, "v_add_f32")
K_OP2(k_sub, "v_sub_f32")
K_OP2(k_max, "v_max_f32")
K_OP2(k_min, "v_min_f32")
K_OP3(k_max3, "v_max3_f32")
K_OP3(k_med3, "v_med3_f32")
K_OP1(k_rcp, "v_rcp_f32")
K_OP1(k_floor, "v_floor_f32")
K_OP1(k_fract, "v_fract_f32")
K_OP1(k_cvtu32, "v_cvt_u32_f32")
K_OP1(k_cvtf32, "v_cvt_f32_u32")
K_OP1(k_ubyte0, "v_cvt_f32_ubyte0")
K_OP1(k_mov, "v_mov_b32")
K_OP2(k_and, "v_and_b32")
K_OP2(k_addu, "v_add_u32")
K_OP2(k_lshl, "v_lshlrev_b32")
K_OP3(k_lshlor, "v_lshl_or_b32")
K_OP3(k_perm, "v_perm_b32")
K_OP3(k_alignbyte, "v_alignbyte_b32")
K_OP3(k_bfe, "v_bfe_u32")
K_OP3(k_bfi, "v_bfi_b32")
K_OP3(k_mad24, "v_mad_u32_u24")
DEFINE_KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
DEFINE_KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %8, %10\n v_cndmask_b32_e64 %1, %1, %8, %10\n v_cndmask_b32_e64 %2, %2, %8, %10\n v_cndmask_b32_e64 %3, %3, %8, %10\n v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %8, %10\n v_cndmask_b32_e64 %6, %6, %8, %10\n v_cndmask_b32_e64 %7, %7, %8, %10\n")
DEFINE_KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n")
DEFINE_KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n")
DEFINE_KERNEL(k_sdwa_cvt, "v_cvt_u32_f32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %2, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %3, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %4, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %5, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %6, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_u32_f32_sdwa %7, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n")
DEFINE_KERNEL(k_mix_fma_perm, "v_fma_f32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")


K_OP2(k_or, "v_or_b32")
K_OP2(k_xor, "v_xor_b32")
K_OP2(k_subu, "v_sub_u32")
K_OP2(k_mul24, "v_mul_u32_u24")
K_OP2(k_ashr, "v_ashrrev_i32")
K_OP3(k_add3, "v_add3_u32")
K_OP3(k_lshladd, "v_lshl_add_u32")
K_OP3(k_andor, "v_and_or_b32")
DEFINE_KERNEL(k_add_clamp, "v_add_f32_e64 %0, %0, %8 clamp\n v_add_f32_e64 %1, %1, %8 clamp\n v_add_f32_e64 %2, %2, %8 clamp\n v_add_f32_e64 %3, %3, %8 clamp\n v_add_f32_e64 %4, %4, %8 clamp\n v_add_f32_e64 %5, %5, %8 clamp\n v_add_f32_e64 %6, %6, %8 clamp\n v_add_f32_e64 %7, %7, %8 clamp\n")
DEFINE_KERNEL(k_sub_abs, "v_sub_f32_e64 %0, %8, |%0|\n v_sub_f32_e64 %1, %8, |%1|\n v_sub_f32_e64 %2, %8, |%2|\n v_sub_f32_e64 %3, %8, |%3|\n v_sub_f32_e64 %4, %8, |%4|\n v_sub_f32_e64 %5, %8, |%5|\n v_sub_f32_e64 %6, %8, |%6|\n v_sub_f32_e64 %7, %8, |%7|\n")
DEFINE_KERNEL(k_mul_e64, "v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8\n")
DEFINE_KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3b808081, %0\n v_mul_f32 %1, 0x3b808081, %1\n v_mul_f32 %2, 0x3b808081, %2\n v_mul_f32 %3, 0x3b808081, %3\n v_mul_f32 %4, 0x3b808081, %4\n v_mul_f32 %5, 0x3b808081, %5\n v_mul_f32 %6, 0x3b808081, %6\n v_mul_f32 %7, 0x3b808081, %7\n")
DEFINE_KERNEL(k_fmac_neg, "v_fmac_f32_e64 %0, -%8, %9\n v_fmac_f32_e64 %1, -%8, %9\n v_fmac_f32_e64 %2, -%8, %9\n v_fmac_f32_e64 %3, -%8, %9\n v_fmac_f32_e64 %4, -%8, %9\n v_fmac_f32_e64 %5, -%8, %9\n v_fmac_f32_e64 %6, -%8, %9\n v_fmac_f32_e64 %7, -%8, %9\n")
DEFINE_KERNEL(k_fmaak, "v_fmaak_f32 %0, %0, %8, 0x3b808081\n v_fmaak_f32 %1, %1, %8, 0x3b808081\n v_fmaak_f32 %2, %2, %8, 0x3b808081\n v_fmaak_f32 %3, %3, %8, 0x3b808081\n v_fmaak_f32 %4, %4, %8, 0x3b808081\n v_fmaak_f32 %5, %5, %8, 0x3b808081\n v_fmaak_f32 %6, %6, %8, 0x3b808081\n v_fmaak_f32 %7, %7, %8, 0x3b808081\n")
DEFINE_KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x3b808081, %8\n v_fmamk_f32 %1, %1, 0x3b808081, %8\n v_fmamk_f32 %2, %2, 0x3b808081, %8\n v_fmamk_f32 %3, %3, 0x3b808081, %8\n v_fmamk_f32 %4, %4, 0x3b808081, %8\n v_fmamk_f32 %5, %5, 0x3b808081, %8\n v_fmamk_f32 %6, %6, 0x3b808081, %8\n v_fmamk_f32 %7, %7, 0x3b808081, %8\n")
DEFINE_KERNEL(k_cvt_sdwa_b1, "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n")
DEFINE_KERNEL(k_max_sdwa, "v_max_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_max_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n")
DEFINE_KERNEL(k_add_sdwa, "v_add_f32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n")

typedef void (*kern_t)(float *, float, float, int);

static void run(const char *name, kern_t k, float *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, 1.0001f, 0.9999f, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, 1.0001f, 0.9999f, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double insts = (double)blocks * threads * iters * 4 * 8;
    printf("%-16s %8.3f ms  %7.2f T lane-inst/s\n", name, best, insts / (best * 1e-3) / 1e12);
}

int main()
{
    float *big; (void)hipMalloc(&big, 256 * 8 * 256 * 4);
#define R(k) run(#k, (kern_t)k, big)
    R(k_fma); R(k_fmac); R(k_mul); R(k_add); R(k_sub); R(k_max); R(k_min); R(k_max3); R(k_med3); R(k_rcp);
    R(k_floor); R(k_fract); R(k_cvtu32); R(k_cvtf32); R(k_ubyte0); R(k_mov); R(k_and); R(k_addu); R(k_lshl);
    R(k_lshlor); R(k_perm); R(k_alignbyte); R(k_bfe); R(k_bfi); R(k_mad24); R(k_cnd_vcc); R(k_cnd_sgpr);
    R(k_cmp_vcc); R(k_cmp_cnd); R(k_sdwa_cvt); R(k_mix_fma_perm); R(k_or); R(k_xor); R(k_subu); R(k_mul24); R(k_ashr); R(k_add3); R(k_lshladd); R(k_andor); R(k_add_clamp); R(k_sub_abs); R(k_mul_e64); R(k_mul_lit); R(k_fmac_neg); R(k_fmaak); R(k_fmamk); R(k_cvt_sdwa_b1); R(k_max_sdwa); R(k_add_sdwa);
    return 0;
}
