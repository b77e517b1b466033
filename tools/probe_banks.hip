// tools/probe_banks.hip -- does the 2-cycle fp32 rate on gfx950 depend on which VGPRs an instruction reads/writes?
// Explicit physical registers: 8 independent instructions per group, dst/src chosen by pattern.
#include <hip/hip_runtime.h>
#include <cstdio>

// INST(d, a, b): one instruction with explicit registers
#define K(NAME, BODY)                                                                              \
    __global__ void NAME(float *out, int iters)                                                    \
    {                                                                                              \
        asm volatile("v_cvt_f32_u32 v40, v0\n v_mov_b32 v41, v40\n v_mov_b32 v42, v40\n v_mov_b32 v43, v40\n"            \
                     "v_mov_b32 v44, v40\n v_mov_b32 v45, v40\n v_mov_b32 v46, v40\n v_mov_b32 v47, v40\n"                \
                     "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 2.0\n v_mov_b32 v51, 4.0\n"               \
                     "v_mov_b32 v52, v40\n v_mov_b32 v53, v40\n v_mov_b32 v54, v40\n v_mov_b32 v55, v40\n"                \
                     "v_mov_b32 v56, v40\n v_mov_b32 v57, v40\n v_mov_b32 v58, v40\n v_mov_b32 v59, v40\n" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59"); \
        for (int i = 0; i < iters; i++) {                                                          \
            asm volatile(BODY BODY BODY BODY ::: "v40","v41","v42","v43","v44","v45","v46","v47","v52","v53","v54","v55","v56","v57","v58","v59"); \
        }                                                                                          \
        float r;                                                                                   \
        asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v52\n v_add_f32 %0, %0, v56" : "=v"(r)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                            \
    }

// in place, src1 constant in bank 0 (v48): like the earlier probes
K(k_inplace_c48, "v_add_f32 v40, v40, v48\n v_add_f32 v41, v41, v48\n v_add_f32 v42, v42, v48\n v_add_f32 v43, v43, v48\n v_add_f32 v44, v44, v48\n v_add_f32 v45, v45, v48\n v_add_f32 v46, v46, v48\n v_add_f32 v47, v47, v48\n")
// dst != src0: d = v52.., a = v40.. (same bank as dst), b const
K(k_dst_other_samebank, "v_add_f32 v52, v40, v48\n v_add_f32 v53, v41, v48\n v_add_f32 v54, v42, v48\n v_add_f32 v55, v43, v48\n v_add_f32 v56, v44, v48\n v_add_f32 v57, v45, v48\n v_add_f32 v58, v46, v48\n v_add_f32 v59, v47, v48\n")
// dst != src0, different bank (d = a+1)
K(k_dst_other_diffbank, "v_add_f32 v53, v40, v48\n v_add_f32 v54, v41, v48\n v_add_f32 v55, v42, v48\n v_add_f32 v56, v43, v48\n v_add_f32 v57, v44, v48\n v_add_f32 v58, v45, v48\n v_add_f32 v59, v46, v48\n v_add_f32 v52, v47, v48\n")
// two variable sources, same bank (a, a+4)
K(k_two_src_samebank, "v_add_f32 v52, v40, v44\n v_add_f32 v53, v41, v45\n v_add_f32 v54, v42, v46\n v_add_f32 v55, v43, v47\n v_add_f32 v56, v44, v40\n v_add_f32 v57, v45, v41\n v_add_f32 v58, v46, v42\n v_add_f32 v59, v47, v43\n")
// two variable sources, different banks (a, a+1)
K(k_two_src_diffbank, "v_add_f32 v52, v40, v41\n v_add_f32 v53, v41, v42\n v_add_f32 v54, v42, v43\n v_add_f32 v55, v43, v44\n v_add_f32 v56, v44, v45\n v_add_f32 v57, v45, v46\n v_add_f32 v58, v46, v47\n v_add_f32 v59, v47, v40\n")
// dependent chain of length 8 (each reads the previous result)
K(k_chain, "v_add_f32 v41, v40, v48\n v_add_f32 v42, v41, v48\n v_add_f32 v43, v42, v48\n v_add_f32 v44, v43, v48\n v_add_f32 v45, v44, v48\n v_add_f32 v46, v45, v48\n v_add_f32 v47, v46, v48\n v_add_f32 v40, v47, v48\n")
// alternate add / mul (different opcodes back to back)
K(k_alt_add_mul, "v_add_f32 v40, v40, v48\n v_mul_f32 v41, v41, v48\n v_add_f32 v42, v42, v48\n v_mul_f32 v43, v43, v48\n v_add_f32 v44, v44, v48\n v_mul_f32 v45, v45, v48\n v_add_f32 v46, v46, v48\n v_mul_f32 v47, v47, v48\n")
// fmac with 3 different registers
K(k_fmac_3reg, "v_fmac_f32 v52, v40, v44\n v_fmac_f32 v53, v41, v45\n v_fmac_f32 v54, v42, v46\n v_fmac_f32 v55, v43, v47\n v_fmac_f32 v56, v44, v40\n v_fmac_f32 v57, v45, v41\n v_fmac_f32 v58, v46, v42\n v_fmac_f32 v59, v47, v43\n")
K(k_fmac_3reg_diffbank, "v_fmac_f32 v52, v41, v46\n v_fmac_f32 v53, v42, v47\n v_fmac_f32 v54, v43, v44\n v_fmac_f32 v55, v40, v45\n v_fmac_f32 v56, v45, v42\n v_fmac_f32 v57, v46, v43\n v_fmac_f32 v58, v47, v40\n v_fmac_f32 v59, v44, v41\n")
// int ops
K(k_and_two_src, "v_and_b32 v52, v40, v41\n v_and_b32 v53, v41, v42\n v_and_b32 v54, v42, v43\n v_and_b32 v55, v43, v44\n v_and_b32 v56, v44, v45\n v_and_b32 v57, v45, v46\n v_and_b32 v58, v46, v47\n v_and_b32 v59, v47, v40\n")
K(k_mul_inline, "v_mul_f32 v52, 0.5, v40\n v_mul_f32 v53, 0.5, v41\n v_mul_f32 v54, 0.5, v42\n v_mul_f32 v55, 0.5, v43\n v_mul_f32 v56, 0.5, v44\n v_mul_f32 v57, 0.5, v45\n v_mul_f32 v58, 0.5, v46\n v_mul_f32 v59, 0.5, v47\n")

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, float *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double ns = best * 1e6 / (8.0 * iters * 32);
    printf("%-24s %8.3f ms  %6.3f ns per wave-inst per SIMD\n", name, best, ns);
}
int main()
{
    float *big; (void)hipMalloc(&big, 256 * 8 * 256 * 4);
#define R(k) run(#k, (kern_t)k, big)
    R(k_inplace_c48); R(k_dst_other_samebank); R(k_dst_other_diffbank); R(k_two_src_samebank); R(k_two_src_diffbank);
    R(k_chain); R(k_alt_add_mul); R(k_fmac_3reg); R(k_fmac_3reg_diffbank); R(k_and_two_src); R(k_mul_inline);
    return 0;
}
