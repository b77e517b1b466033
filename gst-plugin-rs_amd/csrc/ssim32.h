// Internal interface between the C ABI of the SSIM distance (ssim_kernels.hip) and its f32 pipeline (ssim32_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "mi355vfx.h"

namespace mvfx {
namespace ssim32 {

constexpr int kScales = 5;
constexpr int kSlots = 256; // accumulators per (pass, scale): one f64 atomic per workgroup, spread so they do not serialise

// Pass 1 over rows [row_begin, row_end) of the validated pair fr[0] (reference), fr[1]: per-scale sums of the SSIM map + counts;
// the maps stay in this thread's scratch for pass 2.  Synchronises `st`.
int partial_sums(const mvfx_frame *const fr[2], uint32_t row_begin, uint32_t row_end, double sums_out[5], double counts_out[5],
                 uint32_t *n_scales_out, hipStream_t st);
bool pending(); // a partial_sums of this thread is waiting for its partial_deviation
void abandon(); // forget it (the caller starts a pair on the f64 pipeline: the last partial_sums decides which pass 2 runs)
int partial_deviation(const double mean[5], double deviation_sums_out[5], hipStream_t st);

} // namespace ssim32
} // namespace mvfx
