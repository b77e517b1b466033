// tools/probe_isa3.hip -- second census on gfx950: do instruction classes overlap (fast e32 / 4-cycle
// VOP3 / transcendental), and a few more candidate ops for the hsvfilter diet.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_isa3.hip -o /tmp/probe_isa3 && /tmp/probe_isa3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

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
                             : "v"(a), "v"(b), "s"(m), "s"(a)                                  \
                             : "vcc");                                                         \
            }                                                                                  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;    \
    }

#define DEFINE_KERNEL2(NAME, LINE)                                                             \
    __global__ void NAME(float *out, float a, float b, int iters)                             \
    {                                                                                          \
        f2 x0 = {a + threadIdx.x, b}, x1 = {b + threadIdx.x, a}, x2 = {a * 2 + threadIdx.x, b}, \
           x3 = {b * 3 + threadIdx.x, a}, x4 = x0 + 5, x5 = x1 + 6, x6 = x2 + 7, x7 = x3 + 8;  \
        f2 c = {a, b};                                                                         \
        for (int i = 0; i < iters; i++) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; u++)                                      \
            {                                                                                  \
                asm volatile(LINE                                                              \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5),     \
                               "+v"(x6), "+v"(x7)                                              \
                             : "v"(c));                                                        \
            }                                                                                  \
        }                                                                                      \
        f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                \
    }

#define FM(i) "v_fmac_f32 %" #i ", %" #i ", %8\n"
#define PM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define RC(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define CV(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define AD(i) "v_add_f32 %" #i ", %" #i ", %8\n"

DEFINE_KERNEL(k_fmac8, FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7))
DEFINE_KERNEL(k_perm8, PM(0) PM(1) PM(2) PM(3) PM(4) PM(5) PM(6) PM(7))
DEFINE_KERNEL(k_fast4_slow4, FM(0) PM(1) FM(2) PM(3) FM(4) PM(5) FM(6) PM(7))
DEFINE_KERNEL(k_fast6_slow2, FM(0) FM(1) FM(2) PM(3) FM(4) FM(5) FM(6) PM(7))
DEFINE_KERNEL(k_fast7_rcp1, FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) RC(7))
DEFINE_KERNEL(k_fast6_rcp2, FM(0) FM(1) FM(2) RC(3) FM(4) FM(5) FM(6) RC(7))
DEFINE_KERNEL(k_slow7_rcp1, PM(0) PM(1) PM(2) PM(3) PM(4) PM(5) PM(6) RC(7))
DEFINE_KERNEL(k_slow4_rcp4, PM(0) RC(1) PM(2) RC(3) PM(4) RC(5) PM(6) RC(7))
DEFINE_KERNEL(k_cvt4_fast4, CV(0) FM(1) CV(2) FM(3) CV(4) FM(5) CV(6) FM(7))
DEFINE_KERNEL(k_rcp_neg, "v_rcp_f32_e64 %0, -%0\n v_rcp_f32_e64 %1, -%1\n v_rcp_f32_e64 %2, -%2\n v_rcp_f32_e64 %3, -%3\n v_rcp_f32_e64 %4, -%4\n v_rcp_f32_e64 %5, -%5\n v_rcp_f32_e64 %6, -%6\n v_rcp_f32_e64 %7, -%7\n")
#define ML(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8\n"
DEFINE_KERNEL(k_mul_legacy, ML(0) ML(1) ML(2) ML(3) ML(4) ML(5) ML(6) ML(7))
#define PK8(i) "v_cvt_pk_u8_f32 %" #i ", %" #i ", 1, %8\n"
DEFINE_KERNEL(k_cvt_pk_u8, PK8(0) PK8(1) PK8(2) PK8(3) PK8(4) PK8(5) PK8(6) PK8(7))
#define LR(i) "v_lshrrev_b32 %" #i ", 8, %" #i "\n"
DEFINE_KERNEL(k_lshr, LR(0) LR(1) LR(2) LR(3) LR(4) LR(5) LR(6) LR(7))
#define TR(i) "v_trunc_f32 %" #i ", %" #i "\n"
DEFINE_KERNEL(k_trunc, TR(0) TR(1) TR(2) TR(3) TR(4) TR(5) TR(6) TR(7))
#define RN(i) "v_rndne_f32 %" #i ", %" #i "\n"
DEFINE_KERNEL(k_rndne, RN(0) RN(1) RN(2) RN(3) RN(4) RN(5) RN(6) RN(7))
#define SS(i) "v_add_f32 %" #i ", %11, %" #i "\n"
DEFINE_KERNEL(k_add_sgpr, SS(0) SS(1) SS(2) SS(3) SS(4) SS(5) SS(6) SS(7))
#define FS(i) "v_fmac_f32 %" #i ", %11, %" #i "\n"
DEFINE_KERNEL(k_fmac_sgpr, FS(0) FS(1) FS(2) FS(3) FS(4) FS(5) FS(6) FS(7))
#define AI(i) "v_and_b32 %" #i ", 28, %" #i "\n"
DEFINE_KERNEL(k_and_inline, AI(0) AI(1) AI(2) AI(3) AI(4) AI(5) AI(6) AI(7))
#define SI(i) "v_sub_f32 %" #i ", 0.5, %" #i "\n"
DEFINE_KERNEL(k_sub_inline, SI(0) SI(1) SI(2) SI(3) SI(4) SI(5) SI(6) SI(7))
#define F3S(i) "v_fma_f32 %" #i ", %" #i ", %11, %11\n"
DEFINE_KERNEL(k_fma_sgpr, F3S(0) F3S(1) F3S(2) F3S(3) F3S(4) F3S(5) F3S(6) F3S(7))
#define FA(i) "v_fma_f32 %" #i ", -2.0, |%" #i "|, 1.0\n"
DEFINE_KERNEL(k_fma_abs_inline, FA(0) FA(1) FA(2) FA(3) FA(4) FA(5) FA(6) FA(7))
#define MX(i) "v_max_f32 %" #i ", %11, %" #i "\n"
DEFINE_KERNEL(k_max_sgpr, MX(0) MX(1) MX(2) MX(3) MX(4) MX(5) MX(6) MX(7))
#define XO(i) "v_xor_b32 %" #i ", %" #i ", %8\n v_and_b32 %" #i ", %" #i ", %9\n"
DEFINE_KERNEL(k_xor_and_pairs, XO(0) XO(1) XO(2) XO(3))
#define DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_mov_dpp, DPP(0) DPP(1) DPP(2) DPP(3) DPP(4) DPP(5) DPP(6) DPP(7))

#define PKM(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
DEFINE_KERNEL2(k_pk_mul, PKM(0) PKM(1) PKM(2) PKM(3) PKM(4) PKM(5) PKM(6) PKM(7))
#define PKA(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
DEFINE_KERNEL2(k_pk_add, PKA(0) PKA(1) PKA(2) PKA(3) PKA(4) PKA(5) PKA(6) PKA(7))
#define PKF(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n"
DEFINE_KERNEL2(k_pk_fma, PKF(0) PKF(1) PKF(2) PKF(3) PKF(4) PKF(5) PKF(6) PKF(7))
#define PKV(i) "v_pk_mov_b32 %" #i ", %" #i ", %8 op_sel:[0,1]\n"
DEFINE_KERNEL2(k_pk_mov, PKV(0) PKV(1) PKV(2) PKV(3) PKV(4) PKV(5) PKV(6) PKV(7))

typedef void (*kern_t)(float *, float, float, int);

static void run(const char *name, kern_t k, float *dout, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 1000;
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
    printf("%-18s w/simd=%d %8.3f ms  %7.2f T lane-inst/s\n", name, waves_per_simd, best, insts / (best * 1e-3) / 1e12);
}

int main()
{
    float *big; (void)hipMalloc(&big, 256 * 16 * 256 * 4);
#define R(k) run(#k, (kern_t)k, big, 8)
    R(k_fmac8); R(k_perm8); R(k_fast4_slow4); R(k_fast6_slow2); R(k_fast7_rcp1); R(k_fast6_rcp2); R(k_slow7_rcp1);
    R(k_slow4_rcp4); R(k_cvt4_fast4); R(k_rcp_neg); R(k_mul_legacy); R(k_cvt_pk_u8); R(k_lshr); R(k_trunc); R(k_rndne);
    R(k_add_sgpr); R(k_fmac_sgpr); R(k_and_inline); R(k_sub_inline); R(k_fma_sgpr); R(k_fma_abs_inline); R(k_max_sgpr);
    R(k_xor_and_pairs); R(k_mov_dpp); R(k_pk_mul); R(k_pk_add); R(k_pk_fma); R(k_pk_mov);
    run("k_fmac8", (kern_t)k_fmac8, big, 4); run("k_fmac8", (kern_t)k_fmac8, big, 2); run("k_fmac8", (kern_t)k_fmac8, big, 1);
    run("k_perm8", (kern_t)k_perm8, big, 2); run("k_fast4_slow4", (kern_t)k_fast4_slow4, big, 2);
    return 0;
}
