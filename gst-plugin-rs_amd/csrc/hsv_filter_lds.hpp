// Shared by hsv_kernels.hip and hsv_typed_kernels.hip: the LDS selector table of the strength-reduced hsvfilter kernels
// and the launcher of the typed-load kernel.
#pragma once

#include "hsv_math.hpp"
#include "mvfx_internal.h"

namespace mvfx {

constexpr int kHsvBlock = 256;

// LDS tables of one workgroup of the FAST hsvfilter kernels
struct FilterLds {
    uint32_t sextant[8]; // v_perm_b32 selectors, see sextant_selector()
};

// Fills the LDS tables of this workgroup (no-op for the literal variant). blockDim.x == 256.
template <int VARIANT>
__device__ __forceinline__ void init_filter_lds(FilterLds &lds, int off, bool bgr)
{
    if constexpr (VARIANT != kGeneral) {
        if (threadIdx.x < 8)
            lds.sextant[threadIdx.x] = sextant_selector(threadIdx.x, off, bgr);
        __syncthreads();
    }
}

// The settings-dependent part of FastConsts, one per frame of a launch whose frames come from different elements
struct FrameSettings {
    float hue_shift, saturation_mul, saturation_off, value_mul, value_off, neg_saturation_mul;
};
struct FrameSettingsBatch {
    FrameSettings s[kMaxBatch];
};

// hsv_typed_kernels.hip: hsvfilter4_typed_kernel<neg ? kFastNeg : kFast, tile (1 | 2), streaming>
void launch_hsvfilter_typed(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                            uint32_t rows, uint64_t stride, const FastConsts &p, uint32_t word3, uint32_t frame_bytes, int off, bool bgr);

// hsv_typed_kernels.hip: hsvfilter3_typed_kernel (RGB / BGR), same template parameters; word3a / word3b = the two buffer descriptors' format words
void launch_hsvfilter3_typed(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                             uint32_t rows, uint64_t stride, const FastConsts &p, uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, bool bgr);

// hsvfilter3_typed_rows_kernel: RGB / BGR frames whose width is not a multiple of four (always row-padded), the frame's groups as one index space
void launch_hsvfilter3_typed_rows(bool neg_shift, bool streaming, uint32_t n_frames, hipStream_t stream, const FrameBatch &fb, uint32_t width, uint32_t rows,
                                  uint32_t stride, const FastConsts &p, uint32_t word3a, uint32_t word3b, uint32_t frame_bytes, bool bgr);

// hsvfilter4_typed_frames_kernel: the 4-byte kernel with per-frame settings in the kernel arguments (all frames: hue_shift of one sign)
void launch_hsvfilter_typed_frames(bool neg_shift, int tile, bool streaming, dim3 grid, hipStream_t stream, const FrameBatch &fb, uint64_t width,
                                   uint32_t rows, uint64_t stride, const FastConsts &p, const FrameSettingsBatch &fs, uint32_t word3,
                                   uint32_t frame_bytes, int off, bool bgr);

} // namespace mvfx
