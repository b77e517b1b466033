// tools/probe_isa.hip -- gfx950 instruction probes used while designing hsv_math.hpp.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_isa.hip -o /tmp/probe_isa && /tmp/probe_isa
// 1. rounding / saturation of v_cvt_pk_u8_f32
// 2. relative issue cost of v_rcp_f32, v_pk_fma_f32, v_cndmask vs v_fma_f32 (dependent-free streams)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

__global__ void k_cvt(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        unsigned r;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
        out[i] = r;
    }
}

template <int OP>
__global__ void k_rate(float *out, float a, float b, int iters)
{
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 2 + threadIdx.x, x3 = b * 3 + threadIdx.x;
    float y0 = x0 + 1, y1 = x1 + 1, y2 = x2 + 1, y3 = x3 + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (OP == 0) { // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if constexpr (OP == 1) { // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            } else if constexpr (OP == 2) { // v_pk_fma_f32 (2 floats per lane)
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                             : "+v"(*(double *)&x0), "+v"(*(double *)&x2) : "v"(*(double *)&y0), "v"(*(double *)&y2));
            } else if constexpr (OP == 3) { // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
            } else if constexpr (OP == 4) { // v_cmp_lt_f32 -> sgpr pair
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0"
                             :: "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
            } else if constexpr (OP == 5) { // v_cvt_f32_ubyte1
                asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            } else if constexpr (OP == 6) { // v_med3_f32
                asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if constexpr (OP == 7) { // v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if constexpr (OP == 8) { // v_fract_f32
                asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            } else if constexpr (OP == 9) { // v_cvt_pk_u8_f32
                asm volatile("v_cvt_pk_u8_f32 %0, %4, 1, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 1, %2\n v_cvt_pk_u8_f32 %3, %4, 1, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if constexpr (OP == 10) { // v_mul_f32 e32
                asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if constexpr (OP == 11) { // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2"
                             : "+v"(*(double *)&x0), "+v"(*(double *)&x2) : "v"(*(double *)&y0));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int OP>
double time_op(const char *name, float *dout, int lanes_per_inst)
{
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 1.0001f, 0.9999f, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * threads * iters * 8 * 4; // lane-instructions
    double rate = insts / (ms * 1e-3) / 1e12;
    printf("%-18s %8.3f ms  %7.2f T lane-inst/s  (x%d elems/lane-inst = %7.2f T elem-ops/s)\n", name, ms, rate,
           lanes_per_inst, rate * lanes_per_inst);
    return rate;
}

int main()
{
    float vals[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.0f, 255.5f, 300.0f, -0.5f, -1.0f, NAN, INFINITY};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *din; unsigned *dout;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, n * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, n);
    unsigned res[32];
    hipMemcpy(res, dout, n * 4, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32:\n");
    for (int i = 0; i < n; i++) printf("  %10g -> %u\n", vals[i], res[i] & 0xff);

    float *big; hipMalloc(&big, 256 * 8 * 256 * 4);
    time_op<0>("v_fma_f32", big, 1);
    time_op<10>("v_mul_f32", big, 1);
    time_op<1>("v_rcp_f32", big, 1);
    time_op<2>("v_pk_fma_f32", big, 2);
    time_op<11>("v_pk_mul_f32", big, 2);
    time_op<3>("v_cndmask_b32", big, 1);
    time_op<4>("v_cmp_lt_f32", big, 1);
    time_op<5>("v_cvt_f32_ubyte1", big, 1);
    time_op<6>("v_med3_f32", big, 1);
    time_op<7>("v_perm_b32", big, 1);
    time_op<8>("v_fract_f32", big, 1);
    time_op<9>("v_cvt_pk_u8_f32", big, 1);
    return 0;
}
