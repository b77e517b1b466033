// The colorlut kernels of the direct-dispatch lane (csrc/direct_dispatch.h): ONE RGBA8 frame through a 3-D LUT with the two window kernels of
// csrc/colorlut_xwindow.hpp -- the bodies the HIP kernels of colorlut_window_kernels.hip run, instantiated with write-through stores, drained before the
// wave ends (the lane's packets carry no release fence; hsv_direct_kernels.hip has the why).  The stores keep the non-temporal hint on top of the
// write-through (natural-like 4K frames, per-wave windows, no barrier bit: 75.2 k fps without it, 76.4 k with; profiles/r6/colorlut_lane.txt).  Same
// table reads, same arithmetic: same bytes (tests/test_direct_colorlut_gpu.py: all 2^24 colours through both).  A bare code object of its own
// (Makefile: --genco --no-gpu-bundle-output, the flags of colorlut_window_kernels.hip), embedded in the library next to the hsv one; extern "C"
// names, no implicit kernel arguments, 256 lanes per workgroup, grid = (workgroups across, workgroups down).
#include "direct_dispatch_colorlut.h"
#include "colorlut_xwindow.hpp"

extern "C" __global__ __launch_bounds__(256) void mvfx_direct_colorlut_xtile(mvfx::DirectLutArgs a)
{
    mvfx::colorlut_xtile_body<mvfx::kStoreWtNt>(a.in, a.out, a.width, a.height, a.in_stride, a.out_stride, a.p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" __global__ __launch_bounds__(256) void mvfx_direct_colorlut_xwg(mvfx::DirectLutArgs a)
{
    mvfx::colorlut_xwg_body<mvfx::kStoreWtNt>(a.in, a.out, a.width, a.height, a.in_stride, a.out_stride, a.p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
