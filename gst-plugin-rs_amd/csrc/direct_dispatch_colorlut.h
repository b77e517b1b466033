// The direct-dispatch lane (direct_dispatch.h), colorlut's part: one RGBA8 frame through a 3-D LUT with either x-prelerped window kernel.
#pragma once

#include "colorlut_device.hpp"
#include "direct_dispatch.h"

namespace mvfx {

// the ONE kernel argument of the lane's colorlut kernels (csrc/direct/colorlut_direct_kernels.hip)
struct DirectLutArgs {
    const uint8_t *in;
    uint8_t *out;
    uint32_t width, height, in_stride, out_stride; // pixels, pixels, bytes, bytes (rows 16-byte aligned, width a multiple of four: colorlut_impl's tiles_ok)
    LutParams p;                                    // this device's replica of the LUT
};

// Enqueues colorlut on one frame through the lane of the calling thread's current device (wg_window: the workgroup-window kernel, else the per-wave
// windows); the thread's completion event becomes a direct fence.  in_order false (MVFX_OPT_DIRECT_UNORDERED): the packet goes out without the barrier
// bit -- it does not wait for the packets in front of it on its queue.  Return values: direct_hsvfilter_submit.
int direct_colorlut_submit(const DirectLutArgs &args, bool wg_window, int queue, bool in_order);

} // namespace mvfx
