// Host-side tap tables of image 0.25's Lanczos3 resampler (imageops::sample::{vertical,horizontal}_sample):
// computed once per (input size, output size) on the host -- the reference computes the same `ws` vector once per
// output row / column on its streaming thread -- and consumed by the HIP kernels of csrc/imghash_kernels.hip.
#pragma once

#include <cstdint>
#include <vector>

namespace mvfx {

struct LanczosAxis {
    std::vector<uint32_t> left;    // first input sample of every output sample
    std::vector<uint32_t> count;   // number of taps
    std::vector<uint32_t> offset;  // where its weights start in `weights`
    std::vector<float> weights;    // normalised, in tap order
};

// in_size input samples -> out_size output samples (both > 0)
LanczosAxis lanczos3_axis(uint32_t in_size, uint32_t out_size);

} // namespace mvfx
