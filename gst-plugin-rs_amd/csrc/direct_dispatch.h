// The direct-dispatch lane (round 6): one-frame hsvfilter launches as hand-written AQL packets on a queue of the library's own.
//
// What the element's contract asks -- one transform call per buffer (video/hsv/src/hsvfilter/imp.rs:322-326) -- costs a pair of HIP streams 12.3 us per
// 4K frame where a 16-frame launch needs 11.5, and the kernel trace shows no idle gap to blame: two launches overlap completely
// (profiles/r6/single_frame_two_stream_trace.txt).  The loss is per dispatch PACKET: a packet with no work at all, put between the frames, costs
// 0.9-2 us of chip time even on another queue (profiles/r6/kernel_boundary_cost.txt).  Hand-written AQL packets on HSA queues of this process's own,
// same kernel, show which part of the packet it is (tools/probes/aql_scope.cpp): the RELEASE fence every HIP kernel dispatch carries -- an L2
// write-back walk on all eight XCDs when the kernel ends.  Two queues alternating, 4K frames: acquire + release at agent scope (HIP's) 12.2 us per
// frame, release NONE 11.0.  (The barrier bit, the other suspect, is worth nothing on this kernel; HIP's own switch for it, hipExtAnyOrderLaunch, is
// ignored on gfx9 anyway.)  A HIP stream's packets cannot be told to drop their release fence; packets of our own can -- if the kernel's stores need
// none: the lane's kernels store WRITE-THROUGH (sc0 sc1) and drain before they end, which is the write-through publish of MI355X_MICROARCH.md.
// So: per device two HSA queues owned by the library, the frame's kernel goes out with acquire = agent, release = none, its completion signal is
// the frame's fence, its argument block sits in device memory (written through the BAR, as HIP does).  Eligibility and fallbacks:
// include/mi355vfx.h, MVFX_OPT_DIRECT_DISPATCH.
#pragma once

#include <cstdint>

#include "hsv_math.hpp"

namespace mvfx {

// the ONE kernel argument of the lane's kernels (csrc/direct/hsv_direct_kernels.hip): a flat (unpadded) packed 4-byte frame
struct DirectHsvArgs {
    uint8_t *frame;
    uint32_t groups;       // 16-byte pixel groups in the frame = width * height / 4
    uint32_t word3;        // buffer descriptor word 3 of the typed loads (hsvfilter_impl)
    uint32_t frame_bytes;
    int32_t off, bgr;      // colour bytes start at `off`; byte order
    uint32_t reserved;
    FastConsts p;
};

// the argument block of the lane's hsvdetector kernel: flat packed 4-byte frames in (RGBx xRGB BGRx xBGR) and out (RGBA ARGB BGRA ABGR)
struct DirectDetArgs {
    const uint8_t *in;
    uint8_t *out;
    uint32_t groups;       // 16-byte pixel groups = width * height / 4
    uint32_t word3;        // descriptor word 3 of the typed loads (hsvdetector_impl)
    uint32_t in_bytes;
    uint32_t perm_sel;     // v_perm selector that builds the output pixel from the raw input dword and the hit mask
    HsvDetectorParams p;
};

// kernels 0..3: hsvfilter, index = neg_shift * 2 + nontemporal loads; 4: hsvdetector; 5, 6: colorlut, per-wave windows / workgroup window
// (direct_dispatch_colorlut.h)
constexpr int kDirectKernels = 7;
extern const char *const kDirectKernelNames[kDirectKernels];

// The lane has two queues, each IN ORDER (its packets carry the barrier bit): which one a dispatch takes is a function of the `stream` the caller
// passed -- the parity of its index among the calling thread's private streams (mvfx_thread_stream_n), a pointer hash for foreign streams.  The
// elements pick their stream from the buffer's frame number, so the filter and the detector of one frame land on the same queue, one behind the
// other, and consecutive frames alternate.
int direct_queue_hint(hipStream_t stream);

// Enqueues hsvfilter on one flat frame through the lane of the calling thread's current device; the thread's completion event (which must be set)
// becomes a DIRECT fence: it fires when the kernel has finished.  MVFX_OK: enqueued.  1: the lane is not available (no HSA queue, disabled by
// MVFX_DIRECT_DISPATCH=0, the signal could not be made ...) -- nothing was done, the caller launches through its stream.  < 0: MVFX_ERR_*.
int direct_hsvfilter_submit(const DirectHsvArgs &args, bool neg_shift, bool nontemporal, int queue);
int direct_hsvdetector_submit(const DirectDetArgs &args, int queue);

// A dependency ACROSS the lane's queues, waited for on the device: when `e` is a direct fence that has not fired and stands for a dispatch on the
// lane's OTHER queue, a barrier packet with that dispatch's completion signal as its dependency goes into lane queue `queue` (current device) -- the
// next dispatch on `queue` that carries the barrier bit runs behind it; what a HIP stream does with hipStreamWaitEvent, without a host thread waiting.
// Returns 1: the caller's next dispatch on `queue` must keep its barrier bit (a barrier packet went in, or the dispatch `e` stands for sits in `queue`
// itself), and `e` must not be used for ANOTHER dispatch until that next dispatch has finished (its signal would be armed again under the barrier
// packet); 0: nothing to wait for (not a direct fence, or it has fired); < 0: MVFX_ERR_*.
int direct_queue_wait(hipEvent_t e, int queue);

// Every lane dispatch made so far on `device` has FINISHED when this returns (a barrier packet behind each of the lane's queues, waited for on the
// calling thread).  For whoever frees memory a lane kernel may still be reading: hipFree waits for the HIP streams of the device, not for queues it
// does not know (the LUT tables of a mvfx_cube_lut; frame blocks carry their own fence and need none of this).  No lane on the device: returns at once.
void direct_quiesce(int device);

// PARKING.  A process's hardware queues are few: with HIP's four (GPU_MAX_HW_QUEUES) and the lane's two, kernels on HIP streams run at about half their
// speed when several streams are busy -- whether or not the lane's queues carry anything (hsvfilter ! tee ! 2 x hsvdetector on three threads: 19.6 k fps
// without the lane's queues, 10.5 k with them idle; six HIP queues and no lane: 10 k as well; profiles/r6/lane_chain_soak.txt).  So the queues exist only
// while the lane is in use: direct_park drains and destroys them (1: parked now; 0: no lane, or parked already), the next dispatch makes them again.
int direct_park(int device);

// ---- direct fences: an mvfx_event whose last "record" was a lane dispatch ------------------------------------------------------------
// state 0: the event is an ordinary HIP event (or was never used); 1: complete direct fence; 2: pending direct fence
int direct_event_state(hipEvent_t e);
int direct_event_queue(hipEvent_t e);         // the lane queue (0 | 1) of the dispatch a direct fence stands for; -1: not a direct fence
int direct_event_wait(hipEvent_t e);          // host wait; MVFX_OK
void direct_event_forget(hipEvent_t e);       // the event is being recorded / carried the HIP way again
void direct_event_destroy(hipEvent_t e);

} // namespace mvfx
