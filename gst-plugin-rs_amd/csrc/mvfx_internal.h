// Internal helpers shared by the HIP translation units of libmi355vfx (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mi355vfx.h"

namespace mvfx {

// Records the message returned by mvfx_last_error() for this thread and returns `status`.
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Fails with MVFX_ERR_NO_DEVICE when no HIP device is usable; otherwise MVFX_OK.
int require_device();

inline hipStream_t as_stream(mvfx_stream s) { return reinterpret_cast<hipStream_t>(s); }

#define MVFX_HIP_TRY(expr)                                                                    \
    do {                                                                                      \
        hipError_t mvfx_e_ = (expr);                                                          \
        if (mvfx_e_ != hipSuccess)                                                            \
            return ::mvfx::fail(mvfx_e_ == hipErrorNoDevice ? MVFX_ERR_NO_DEVICE              \
                                                            : MVFX_ERR_DEVICE,                \
                                "%s failed: %s", #expr, hipGetErrorString(mvfx_e_));          \
    } while (0)

// pixel_stride()[0] of the packed formats; 0 for planar / unknown
inline int bytes_per_pixel(int format)
{
    switch (format) {
    case MVFX_FORMAT_RGBX: case MVFX_FORMAT_XRGB: case MVFX_FORMAT_BGRX: case MVFX_FORMAT_XBGR:
    case MVFX_FORMAT_RGBA: case MVFX_FORMAT_ARGB: case MVFX_FORMAT_BGRA: case MVFX_FORMAT_ABGR:
    case MVFX_FORMAT_RGB10A2_LE:
        return 4;
    case MVFX_FORMAT_RGB: case MVFX_FORMAT_BGR:
        return 3;
    case MVFX_FORMAT_RGBA64_LE: case MVFX_FORMAT_RGBA64_BE:
        return 8;
    default:
        return 0;
    }
}

// Basic validation shared by every packed-frame entry point.
int check_packed_frame(const mvfx_frame *f, const char *what);

// Kernel options of the calling thread (mvfx_thread_set_options); the thread is the library's context.
uint32_t thread_options();
inline int opt_hsv_variant() { const uint32_t o = thread_options(); return (o & MVFX_OPT_HSV_LITERAL) ? 1 : ((o & MVFX_OPT_HSV_FORCE_FAST) ? 2 : 0); }
inline bool opt_nontemporal() { return (thread_options() & MVFX_OPT_NONTEMPORAL) != 0; }
inline bool opt_typed_loads() { return (thread_options() & MVFX_OPT_HSV_VALU_UNORM) == 0; }
inline bool opt_direct() { return (thread_options() & MVFX_OPT_DIRECT_DISPATCH) != 0; }
inline bool opt_direct_only() { return (thread_options() & (MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY)) == (MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY); }
inline bool opt_direct_unordered() { return (thread_options() & MVFX_OPT_DIRECT_UNORDERED) != 0; }
inline int opt_lut_placement() { return (int)((thread_options() & MVFX_OPT_LUT_PLACEMENT_MASK) >> MVFX_OPT_LUT_PLACEMENT_SHIFT); }

// Grow-only device scratch used by the *_host entry points (one per thread and device).
int host_scratch(size_t bytes, int slot, void **out);
hipStream_t host_stream();
hipStream_t host_stream_n(uint32_t index); // 0 = host_stream(), 1..3 = further private streams of the calling thread
int thread_stream_index(hipStream_t stream); // 0..3: `stream` is that private stream of the calling thread (current device); -1: it is not
// Grow-only device scratch keyed by (thread, device, stream): intermediate results handed from one launch to the next on `stream`.
int stream_scratch(hipStream_t stream, size_t bytes, void **out);

// The completion event of the calling thread (mvfx_thread_set_completion_event): while one is set, every kernel this thread launches
// carries it as the STOP EVENT of its own dispatch packet (hipExtLaunchKernelGGL) -- the fence of the element layer without a barrier
// packet behind the kernel: hipEventRecord behind every 4K launch costs 2.6 us of device time, the attached event nothing
// (tools/probes/event_cost.hip: 78.4 k against 97.9 k launches/s, 98.5 k with no fence at all).  Several launches of one call
// re-record it; the last one stands.
hipEvent_t completion_event();
void note_completion_event_used();
#define MVFX_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                              \
    do {                                                                                                                   \
        hipEvent_t mvfx_done_ = ::mvfx::completion_event();                                                                \
        if (mvfx_done_) {                                                                                                  \
            ::mvfx::note_completion_event_used();                                                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, mvfx_done_, 0, __VA_ARGS__);                 \
        } else {                                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                           \
        }                                                                                                                  \
    } while (0)

constexpr int kMaxBatch = 32; // frames per launch of the batched entry points

struct FrameBatch {
    uint8_t *base[kMaxBatch];
};

} // namespace mvfx
