/*
 * mi355vfx.h -- C ABI of the MI355X-native per-pixel video-filter kernels.
 *
 * This is the drop-in boundary for the `VideoFilterImpl::transform_frame*` hot loops of
 * sdroege/gst-plugin-rs `video/hsv`, `video/colorlut` and `video/videofx`.  Every entry
 * point names the reference loop it replaces (file:line under the reference tree).  A Rust
 * `imp.rs` (or the C++ element layer in gst-plugin-rs_amd/host) maps the GstVideoFrame,
 * fills an `mvfx_frame` from `plane_data(0)` / `plane_stride()[0]` / `width()` / `height()` /
 * `format()` and calls one function; see INTEGRATION.md for the `extern "C"` block.
 *
 * Conventions
 *   - plain C, no GLib/GStreamer/torch types; all structs are POD with fixed-width fields.
 *   - every function returns MVFX_OK (0) or a negative mvfx_status; the message of the last
 *     failure on the calling thread is available from mvfx_last_error().
 *   - `*_device` style entry points (no suffix) take DEVICE pointers and are asynchronous on
 *     `stream` (a hipStream_t cast to void*, NULL = null stream).  The caller owns all frame
 *     memory.  `*_host` entry points take HOST pointers (a mapped GstBuffer), stage through
 *     library-owned device scratch and return after the result is back in host memory, which
 *     is what a GstVideoFilter vfunc needs (frames are borrowed, SURVEY.md 8b "Ownership").
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     MVFX_ERR_NO_DEVICE / MVFX_ERR_DEVICE.
 */
#ifndef MI355VFX_H
#define MI355VFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVFX_ABI_VERSION 2

/* Pixel formats on the path (GstVideoFormat names).  hsvfilter: the first ten
 * (hsvfilter/imp.rs:278-289); hsvdetector sink RGBx..BGR, src RGBA..ABGR
 * (hsvdetector/imp.rs:78-96); colorlut RGBA, RGBA64_LE/BE (colorlut/imp.rs:122-134);
 * colordetect RGB,RGBA,ARGB,BGR,BGRA (colordetect/imp.rs:214-221); videocompare RGB,RGBA;
 * colorlut additionally RGB10A2_LE on device memory (the D3D12 variant's third format, d3d12colorlut/imp.rs:236-244);
 * roundedcorners I420 -> A420 (border/imp.rs:345-365). */
typedef enum mvfx_format {
    MVFX_FORMAT_RGBX = 0,
    MVFX_FORMAT_XRGB = 1,
    MVFX_FORMAT_BGRX = 2,
    MVFX_FORMAT_XBGR = 3,
    MVFX_FORMAT_RGBA = 4,
    MVFX_FORMAT_ARGB = 5,
    MVFX_FORMAT_BGRA = 6,
    MVFX_FORMAT_ABGR = 7,
    MVFX_FORMAT_RGB = 8,
    MVFX_FORMAT_BGR = 9,
    MVFX_FORMAT_RGBA64_LE = 10,
    MVFX_FORMAT_RGBA64_BE = 11,
    MVFX_FORMAT_I420 = 12,
    MVFX_FORMAT_A420 = 13,
    MVFX_FORMAT_RGB10A2_LE = 14, /* colorlut only: third format of d3d12colorlut's caps (d3d12colorlut/imp.rs:236-244) */
    MVFX_FORMAT_NV12 = 15        /* converters only: Y plane + one plane of interleaved (U, V) pairs */
} mvfx_format;

typedef enum mvfx_status {
    MVFX_OK = 0,
    MVFX_ERR_INVALID_ARGUMENT = -1,  /* NULL pointer, zero stride, row longer than stride ... */
    MVFX_ERR_UNSUPPORTED_FORMAT = -2,/* format not in the element's caps (reference: unreachable!()) */
    MVFX_ERR_NOT_NEGOTIATED = -3,    /* frame sizes differ (videocompare/imp.rs:337-346) */
    MVFX_ERR_DEVICE = -4,            /* a HIP call failed; maps to GST_FLOW_ERROR */
    MVFX_ERR_NO_DEVICE = -5,         /* no HIP device: the product has no CPU fallback */
    MVFX_ERR_PARSE = -6,             /* .cube text rejected (parser.rs CubeParseError::InvalidLut) */
    MVFX_ERR_IO = -7,                /* .cube file unreadable (CubeParseError::Io) */
    MVFX_ERR_REFERENCE_PANIC = -8,   /* input on which the reference panics (assert_eq!, slice range) */
    MVFX_ERR_NO_LUT = -9,            /* colorlut without a parsed LUT (colorlut/imp.rs:209-213) */
    MVFX_ERR_OUT_OF_MEMORY = -10,
    MVFX_ERR_DIRECT_UNAVAILABLE = -11 /* MVFX_OPT_DIRECT_ONLY: the direct-dispatch lane cannot take this frame; nothing was launched */
} mvfx_status;

/* hipStream_t passed through as an opaque pointer; NULL selects the null stream. */
typedef void *mvfx_stream;

/* One mapped plane-0 view of a packed frame == what the reference reads from
 * gst_video::VideoFrameRef: plane_data(0) (stride*height bytes), plane_stride()[0],
 * width(), height(), format().  Planar I420/A420 frames use mvfx_planar_frame. */
typedef struct mvfx_frame {
    void *data;      /* first byte of plane 0 */
    uint32_t width;  /* pixels */
    uint32_t height; /* rows; plane_data(0).len() == stride * height */
    uint32_t stride; /* bytes between rows */
    int32_t format;  /* mvfx_format */
} mvfx_frame;

/* ---- library / device ---- */
int mvfx_abi_version(void);
const char *mvfx_last_error(void);        /* thread-local, never NULL */
const char *mvfx_status_string(int status);
int mvfx_device_count(void);              /* 0 when no HIP device is visible */
int mvfx_set_device(int ordinal);
int mvfx_current_device(void);            /* ordinal of the calling thread's current device, -1 without a device */
int mvfx_stream_synchronize(mvfx_stream stream);

/* Device buffers for callers that do not bring their own allocator (tests, the element
 * layer's staging, bench).  Plain hipMalloc/hipFree/hipMemcpy underneath. */
int mvfx_device_alloc(void **out_ptr, size_t bytes);
int mvfx_device_free(void *ptr);
int mvfx_copy_to_device(void *dst_device, const void *src_host, size_t bytes, mvfx_stream stream);
int mvfx_copy_to_host(void *dst_host, const void *src_device, size_t bytes, mvfx_stream stream);
int mvfx_copy_device_to_device(void *dst_device, const void *src_device, size_t bytes, mvfx_stream stream);
/* The calling thread's private non-blocking stream (the one the *_host entry points use); lets an
 * element layer issue the device entry points of one streaming thread in order. */
mvfx_stream mvfx_thread_stream(void);
/* Further private streams of the calling thread (index 0 = mvfx_thread_stream(), 1..3 more, index taken modulo 4).  Consecutive
 * buffers of one video stream are independent frames: an element that alternates between two of these per buffer lets the tail of
 * one frame's kernel overlap the head of the next one's (a launch per 4K frame: 16.2 us back to back on ONE stream); the buffers'
 * fences keep every consumer correct. */
mvfx_stream mvfx_thread_stream_n(uint32_t index);

/* ---- events: the fence a device-memory element leaves on its output instead of blocking ----
 * What the reference's d3d12colorlut does with ID3D12Fence (set a fence value on the output memory and return,
 * d3d12colorlut/imp.rs:695-714): the producer records an event on its stream after the last kernel that touches a
 * buffer, the next user makes ITS stream wait for it (device-side wait, the host never blocks) and a CPU map waits on
 * the host.  hipEvent_t passed through as an opaque pointer. */
typedef void *mvfx_event;
int mvfx_event_create(mvfx_event *out);
int mvfx_event_destroy(mvfx_event event);
int mvfx_event_record(mvfx_event event, mvfx_stream stream);
int mvfx_stream_wait_event(mvfx_stream stream, mvfx_event event);
int mvfx_event_synchronize(mvfx_event event);
/* 1: everything recorded before the event has finished; 0: still running (hipEventQuery -> hipErrorNotReady); < 0: MVFX_ERR_*.  Never blocks. */
int mvfx_event_query(mvfx_event event);
/* 1: the event's last "record" was a direct dispatch (MVFX_OPT_DIRECT_DISPATCH): no stream is ordered behind the work it stands for.  0: an ordinary
 * event (recorded on a stream, or carried by a kernel of a stream as its stop event), or never used. */
int mvfx_event_is_direct(mvfx_event event);
/* The lane queue (0 or 1) of the dispatch a direct fence stands for, -1 when the event is not a direct fence; and the queue a call with
 * MVFX_OPT_DIRECT_DISPATCH and this `stream` takes: the parity of the stream's index among the calling thread's private streams
 * (mvfx_thread_stream_n), a pointer hash for other streams.  Two dispatches on one queue run in the order they were made. */
int mvfx_event_direct_queue(mvfx_event event);
int mvfx_direct_queue_of_stream(mvfx_stream stream);
/* A dependency across the lane's two queues, waited for ON THE DEVICE (what mvfx_stream_wait_event is to HIP streams): when `event` is a direct fence
 * that has not fired, the caller's NEXT direct dispatch on lane queue `queue` (current device) runs behind the dispatch the fence stands for -- a
 * barrier packet with its completion signal goes into `queue` when that dispatch sits in the other queue.  1: armed -- the next dispatch on `queue` must
 * be an in-order one (not MVFX_OPT_DIRECT_UNORDERED), and `event` must not be set as the completion event of another call until that dispatch has
 * finished; 0: nothing to wait for (an ordinary event, or the fence has fired); < 0: MVFX_ERR_*. */
int mvfx_direct_queue_wait_event(int queue, mvfx_event event);
/* The lane's two hardware queues of the current device are drained and destroyed (1), or there were none (0); the next direct dispatch makes them again.
 * For a caller that sees its frames go to streams anyway: hardware queues are few, and two idle ones beside HIP's four slow kernels on busy HIP streams
 * down to half (csrc/direct_dispatch.h, "PARKING"). */
int mvfx_direct_lane_park(void);
/* The fence without a barrier packet.  While a completion event is set on the calling thread, every kernel the thread launches
 * through this library carries it as the stop event of its own dispatch (hipExtLaunchKernelGGL): the event is recorded when the
 * kernel finishes, with no packet of its own behind it -- hipEventRecord behind every 4K launch costs 2.6 us of device time
 * (tools/probes/event_cost.hip).  Usage: set, make ONE library call, clear.  _clear returns how many launches carried the event:
 * 0 means the call launched nothing (or took a path that does not launch kernels) and the caller records the event itself.
 * Several launches of one call re-record the event; the last one stands, which is the fence of the whole call on an in-order stream. */
int mvfx_thread_set_completion_event(mvfx_event event);
int mvfx_thread_clear_completion_event(void);

/* Page-locked host memory (hipHostMalloc) for upload / download staging: a copy from or to it is a real DMA at PCIe
 * speed instead of the runtime's chunked staging of pageable memory; and copies that do NOT synchronise (the caller
 * orders them with the stream / an event; the host block must stay valid until then). */
int mvfx_host_alloc(void **out_ptr, size_t bytes);
int mvfx_host_free(void *ptr);
int mvfx_copy_to_device_async(void *dst_device, const void *src_host, size_t bytes, mvfx_stream stream);
int mvfx_copy_to_host_async(void *dst_host, const void *src_device, size_t bytes, mvfx_stream stream);
int mvfx_copy_device_to_device_async(void *dst_device, const void *src_device, size_t bytes, mvfx_stream stream);

/* ---- per-thread kernel options ----
 * The calling thread is the library's implicit context: its private stream (mvfx_thread_stream), its staging scratch
 * and these options are thread-local, so elements on different streaming threads never see each other's choice and
 * an element that shares a thread with others sets its word before its call (one TLS store).  Every combination
 * produces the same bytes; the options pick cache policy and kernel variant.  0 (default) = automatic. */
#define MVFX_OPT_NONTEMPORAL 0x01u     /* hsvfilter, hsvdetector: the OUTPUT leaves the GPU or is not read again before ~256 MB of other traffic -- the
                                          typed kernels store it WRITE-THROUGH with the non-temporal hint (`sc0 sc1 nt`; round 6, csrc/device_store.hpp):
                                          nothing stays dirty in the L2s, the release fence at the end of the dispatch has nothing to write back (16 x 4K
                                          per launch +2-5 %, one frame per call on a stream 81 k -> 90.7 k fps).  Leave clear when the next kernel reads
                                          the frame (hsvfilter ! hsvdetector on 1080p: 3 % for the pair); the VALU kernels take it as non-temporal
                                          loads / stores, as rounds 1-5 did */
#define MVFX_OPT_HSV_LITERAL 0x02u     /* hsvfilter/hsvdetector: force the literal transcription (IEEE divides, fmodf) */
#define MVFX_OPT_HSV_FORCE_FAST 0x04u  /* force the strength-reduced kernels; MVFX_ERR_INVALID_ARGUMENT when the settings
                                          are outside their proven domain instead of silently running the literal ones */
#define MVFX_OPT_HSV_VALU_UNORM 0x08u  /* `byte / 255.0` on the VALU instead of typed buffer loads (texture-unit UNORM8
                                          conversion, exact for all 256 byte values: tools/probe_unorm.hip) */
#define MVFX_OPT_LUT_PLACEMENT_SHIFT 4 /* colorlut LUT placement: 0 auto | 1 node layout in global/L2 | 2 LDS |         */
#define MVFX_OPT_LUT_PLACEMENT_MASK 0x70u /* 3 cell-packed global | 4 literal kernels | 5 tile kernel (wave-local LUT     */
                                          /* window in LDS, the automatic choice for 3-D cubes) | 6 baked table (RGBA8 only)  */
                                          /* | 7 round 4's per-wave x-prelerped windows instead of the workgroup window (A/B); */
                                          /* (placement << SHIFT) & MASK.  5 and 7 are preferences, not demands: frames the */
                                          /* window kernels cannot take (1-D LUTs, cubes above 65 points, non-finite domains, */
                                          /* widths that are not multiples of 4, rows that are not 16-byte aligned) run the */
                                          /* kernel the automatic choice would have picked for them -- same bytes either way */
#define MVFX_OPT_SSIM_F64 0x80u        /* hash-algo=dssim: f64 planes and window sums (round 2's pipeline, within 1e-9 of the f64
                                          checker) instead of the default f32 pipeline (what dssim-core computes in) */
#define MVFX_OPT_LUT_WG_WINDOW 0x100u  /* colorlut, placement 0 on RGBA8 frames and cubes of 5+ points: always the workgroup-window kernel
                                          (by default a content probe of an earlier frame of the LUT's stream chooses between it -- busy
                                          pictures -- and the per-wave windows of placement 7 -- calm ones; same bytes either way) */
#define MVFX_OPT_DIRECT_DISPATCH 0x200u /* (colorlut: see below) hsvfilter, ONE frame per call (mvfx_hsvfilter_transform_frame_ip), packed 4-byte formats without row padding, settings in
                                          the strength-reduced kernels' domain, a completion event set on the thread: the library may enqueue the frame's
                                          kernel on a queue of ITS OWN instead of `stream` -- a hand-written AQL packet without the release fence every
                                          kernel dispatch of a HIP stream carries (an L2 write-back walk on eight XCDs; the lane's kernels store
                                          write-through instead): one 4K frame per call 12.3 us on two alternating streams, 11.0 this way
                                          (csrc/direct_dispatch.h).  Also mvfx_hsvdetector_transform_frame, 4-byte input formats.  What the caller promises
                                          by setting the bit: (1) everything the frame depends on has FINISHED (nothing of it is merely enqueued on `stream`)
                                          or is itself a direct dispatch on the lane queue this call takes -- mvfx_direct_queue_of_stream(stream); the lane's
                                          two queues are each in order --, (2) it takes the frame's completion from the thread's completion event only, NOT
                                          from the order of `stream`.
                                          That event is then a DIRECT fence: mvfx_event_query / _synchronize work as ever; mvfx_stream_wait_event returns at
                                          once when it has fired and otherwise makes the calling THREAD wait (a HIP stream cannot wait for it on the device).
                                          A frame or a box the lane cannot take (row padding, RGB / BGR, literal-kernel settings, MVFX_DIRECT_DISPATCH=0, no
                                          HSA queue) is launched on `stream` as if the bit were clear; mvfx_event_is_direct says which it was.  Same bytes.
                                          Also mvfx_colorlut_transform_frame: ONE RGBA8 frame pair through a 3-D LUT of 4+ points with the window kernels
                                          (placement 0 or 7; width a multiple of four, rows 16-byte aligned -- row padding is fine here).  In queue order
                                          the lane buys colorlut little (its kernel is not bound by the release fence); see MVFX_OPT_DIRECT_UNORDERED. */
#define MVFX_OPT_DIRECT_ONLY 0x400u     /* with MVFX_OPT_DIRECT_DISPATCH: a frame the lane cannot take is NOT launched on `stream`; the call returns
                                          MVFX_ERR_DIRECT_UNAVAILABLE and has done nothing (a caller whose promise (1) rests on the lane's queue order must
                                          not be moved to a stream behind its back) */
#define MVFX_OPT_DIRECT_UNORDERED 0x800u /* with MVFX_OPT_DIRECT_DISPATCH, colorlut only (the hsv kernels gain nothing from it): the caller promises that everything
                                          the frame depends on has FINISHED -- promise (1) without its second half -- and the packet goes out WITHOUT the
                                          barrier bit: it may start while earlier packets of its lane queue still run, as the frames of a batched launch do
                                          (one 4K frame per call through a 33^3 LUT, natural-like content: 66-70 k fps on two streams, 70-72 k on the lane in
                                          order, 75-76 k this way; 16 frames per launch: 81 k).  Later packets WITH the bit still wait for it. */
int mvfx_thread_set_options(uint32_t options);
uint32_t mvfx_thread_options(void);

/* ---- hsvfilter : video/hsv/src/hsvfilter/imp.rs ----
 * Settings == `struct Settings` hsvfilter/imp.rs:32-39 (defaults :25-29: 0,1,0,1,0). */
typedef struct mvfx_hsvfilter_settings {
    float hue_shift;
    float saturation_mul;
    float saturation_off;
    float value_mul;
    float value_off;
} mvfx_hsvfilter_settings;

/* Replaces HsvFilter::transform_frame_ip + hsv_filter (hsvfilter/imp.rs:322-377, :76-120):
 * in place, per pixel from_rgb|from_bgr -> hue shift / sat, val affine+clamp -> to_rgb|to_bgr;
 * the 4th byte of 4-byte formats and all row padding are left untouched.
 * Formats: RGBx xRGB BGRx xBGR RGBA ARGB BGRA ABGR RGB BGR.  Device memory, async. */
int mvfx_hsvfilter_transform_frame_ip(const mvfx_frame *frame,
                                      const mvfx_hsvfilter_settings *settings,
                                      mvfx_stream stream);

/* Same loop over `n_frames` independent frames of identical geometry and format (e.g. one
 * frame from each of N streams) in a single launch: amortises the ~1.5 us kernel boundary
 * that a 12 us 4K frame would otherwise pay per buffer. */
int mvfx_hsvfilter_transform_frames_ip(const mvfx_frame *frames, uint32_t n_frames,
                                       const mvfx_hsvfilter_settings *settings,
                                       mvfx_stream stream);

/* The same single launch with the settings of EVERY frame given (settings[i] for frames[i]): frames of different hsvfilter
 * elements.  Frames whose hue-shift has the same sign share a launch (the per-frame values travel in the kernel arguments). */
int mvfx_hsvfilter_transform_frames_ip_settings(const mvfx_frame *frames, uint32_t n_frames,
                                                const mvfx_hsvfilter_settings *settings, mvfx_stream stream);

/* Launch combiner: same contract as mvfx_hsvfilter_transform_frame_ip -- one call per buffer (hsvfilter/imp.rs:322-326),
 * asynchronous, ordered behind what the caller enqueued on `stream` before and ahead of what it enqueues afterwards -- but
 * the frames that the threads of one process submit at about the same time leave as ONE batched launch (per device; up to 16
 * frames, each with its own settings).  No extra thread: the first caller of a batch leads it -- it waits at most
 * MVFX_COMBINE_WINDOW_US (environment, default 30) microseconds for the other recently active streams' frames and launches all of
 * them on its own stream -- the others follow (their streams wait for the batch's event).  A call returns as soon as its frame's
 * launch has been enqueued; the host never waits for the GPU; a lone stream is never held back.  Measured numbers: DESIGN.md. */
int mvfx_hsvfilter_transform_frame_ip_combined(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings,
                                               mvfx_stream stream);
/* The combiner without caller streams: the frame's ordering is given as events.  `wait_for` (may be NULL) is the fence its previous
 * user left on the frame; `*done_out` receives the event behind the launch that filters it -- the frame's new fence, owned by the
 * library, valid for the life of the process (it may come to stand for a LATER launch of the same in-order stream: waiting for it
 * is then conservative, never wrong).  All fenced launches of a device run on one library-owned stream in submission order, so
 * consecutive batches are separated by a kernel boundary only -- no cross-stream waits on either side, which is what made the
 * stream-ordered variant above slower than per-stream launches.  This is the entry the element uses with MVFX_COMBINE=2. */
int mvfx_hsvfilter_transform_frame_ip_fenced(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings,
                                             mvfx_event wait_for, mvfx_event *done_out);
/* batches launched and frames carried by the combiner of `device` so far (frames / batches = the average batch) */
int mvfx_combiner_stats(int device, uint64_t *batches_out, uint64_t *frames_out);
/* average time a combined call took on the host (collection wait + launch or event hand-over), in us;
 * MVFX_COMBINE_STATS=1 in the environment prints all three numbers to stderr at process exit (gst-launch runs) */
double mvfx_combiner_average_wait_us(int device);

/* Host-memory variant for a GstVideoFilter vfunc working on system-memory buffers:
 * H2D -> kernel -> D2H on an internal stream, returns when `frame->data` holds the result. */
int mvfx_hsvfilter_transform_frame_ip_host(const mvfx_frame *frame,
                                           const mvfx_hsvfilter_settings *settings);

/* ---- hsvdetector : video/hsv/src/hsvdetector/imp.rs ----
 * Settings == `struct Settings` hsvdetector/imp.rs:34-42 (defaults :26-31). */
typedef struct mvfx_hsvdetector_settings {
    float hue_ref;
    float hue_var;
    float saturation_ref;
    float saturation_var;
    float value_ref;
    float value_var;
} mvfx_hsvdetector_settings;

/* Replaces HsvDetector::transform_frame + hsv_detect (hsvdetector/imp.rs:422-707, :100-160):
 * out of place; input RGBx xRGB BGRx xBGR RGB BGR, output RGBA ARGB BGRA ABGR; the colour is
 * copied in the output order and alpha is 255 where the pixel's HSV lies within
 * ref +- var (hue circular), else 0.  in/out must have equal width and height. */
int mvfx_hsvdetector_transform_frame(const mvfx_frame *in_frame, const mvfx_frame *out_frame,
                                     const mvfx_hsvdetector_settings *settings,
                                     mvfx_stream stream);
/* n_frames independent streams in one launch (frame i of in_frames -> frame i of out_frames; all
 * pairs share geometry and formats; <= 32 pairs per launch, more are split).  One 1080p frame is
 * only ~6 us of GPU work, so single-frame launches are launch-bound; this is the multi-stream form. */
int mvfx_hsvdetector_transform_frames(const mvfx_frame *in_frames, const mvfx_frame *out_frames,
                                      uint32_t n_frames, const mvfx_hsvdetector_settings *settings,
                                      mvfx_stream stream);
int mvfx_hsvdetector_transform_frame_host(const mvfx_frame *in_frame, const mvfx_frame *out_frame,
                                          const mvfx_hsvdetector_settings *settings);

/* f32 HSV of every pixel of a packed 4-byte frame (hsvutils::from_rgb / from_bgr,
 * hsvutils.rs:44-128) as 3 floats per pixel, row-major without padding.  Exists so the
 * "within 1 ULP on the f32 HSV path" claim can be checked directly (tests only need it;
 * an element never calls it). */
int mvfx_hsv_from_frame(const mvfx_frame *frame, float *hsv_out_device, mvfx_stream stream);

/* Diagnostic: checks on the current device the hardware property the RGB / BGR (3-byte) hsvfilter and hsvdetector kernels rest on --
 * a typed buffer load (DATA_FORMAT 8_8_8_8, NUM_FORMAT UNORM) at a byte address that is NOT a multiple of four delivers
 * RN(byte / 255.0f) of the bytes AT that address, through both descriptors and the four immediate offsets the kernels use, for every
 * byte value at every address alignment, in RGB and BGR channel order (`let r = in_p[0] as f32 / 255.0`, hsvutils.rs:45-55, done
 * by the texture unit).  *checked_out = comparisons made (24 576), *mismatches_out = how many differed bit-wise from the IEEE
 * division.  Synchronous; tests call it, an element never does. */
int mvfx_selftest_typed_unorm8(uint32_t *checked_out, uint32_t *mismatches_out);

/* ---- colorlut : video/colorlut/src/parser.rs + colorlut/imp.rs ----
 * mvfx_cube_lut mirrors `CubeLut` (parser.rs:68-74): domain_scale/offset + 1-D tables or the
 * 3-D [r,g,b,1.0] node array, R fastest.  Parsing is host-only (works without a GPU); the device
 * copy is made lazily by the first transform on each device.  The handle is owned by the
 * caller: create in `start()` (colorlut/imp.rs:168-194), free in `stop()` (:196-199). */
typedef struct mvfx_cube_lut mvfx_cube_lut;

/* Replaces CubeLut::parse (parser.rs:110-282).  MVFX_ERR_PARSE + message on rejection. */
int mvfx_cube_lut_parse(const char *text, size_t len, mvfx_cube_lut **out);
/* Replaces CubeLut::parse_file (parser.rs:105-108).  MVFX_ERR_IO when unreadable / not UTF-8. */
int mvfx_cube_lut_parse_file(const char *path, mvfx_cube_lut **out);
void mvfx_cube_lut_free(mvfx_cube_lut *lut);
int mvfx_cube_lut_is_3d(const mvfx_cube_lut *lut);
uint32_t mvfx_cube_lut_size(const mvfx_cube_lut *lut);
/* Diagnostic: on how many devices the handle holds a device copy.  The copy of a device is made by the first transform a thread makes there and is
 * shared by every later user of the handle on that device; a handle used from streaming threads on several GPUs keeps one per GPU (the reference's
 * d3d12colorlut rebuilds its context when the device of the incoming memory changes, d3d12colorlut/imp.rs:494-542) until mvfx_cube_lut_free. */
int mvfx_cube_lut_device_copies(const mvfx_cube_lut *lut);
/* Diagnostic: the last verdict of the LUT's content probe (see MVFX_OPT_LUT_WG_WINDOW) -- 0 none yet, 1 calm, 2 busy -- and, when
 * `busy_blocks` is not NULL, how many of the 256 sampled blocks of the probed frame were busy.  Never synchronises. */
int mvfx_cube_lut_content_verdict(const mvfx_cube_lut *lut, uint32_t *busy_blocks);
int mvfx_cube_lut_domain(const mvfx_cube_lut *lut, float scale[3], float offset[3]);
const float *mvfx_cube_lut_rgba(const mvfx_cube_lut *lut);               /* host, size^3*4 or NULL */
const float *mvfx_cube_lut_table_1d(const mvfx_cube_lut *lut, int channel); /* host, size or NULL */

/* Replaces ColorLut::transform_frame -> transform_rgba{,64}_{1d,3d} (colorlut/imp.rs:203-397):
 * out of place, RGBA / RGBA64_LE / RGBA64_BE (in and out the same format), alpha copied.
 * lut == NULL -> MVFX_ERR_NO_LUT (FlowError::Error, colorlut/imp.rs:209-213). */
int mvfx_colorlut_transform_frame(mvfx_cube_lut *lut, const mvfx_frame *in_frame,
                                  const mvfx_frame *out_frame, mvfx_stream stream);
/* n_frames frames (e.g. one of each of n streams graded with the same LUT) in one launch; all pairs share
 * geometry and format; <= 32 pairs per launch, more are split. */
int mvfx_colorlut_transform_frames(mvfx_cube_lut *lut, const mvfx_frame *in_frames,
                                   const mvfx_frame *out_frames, uint32_t n_frames,
                                   mvfx_stream stream);
int mvfx_colorlut_transform_frame_host(mvfx_cube_lut *lut, const mvfx_frame *in_frame,
                                       const mvfx_frame *out_frame);
/* Where the LUT is read from / which kernel runs is a thread option (MVFX_OPT_LUT_PLACEMENT_*, see
 * mvfx_thread_set_options): 0 = automatic (3-D cubes of 3..65 points on 16-byte aligned frames whose width is a multiple of
 * 4: RGBA8 on cubes of 4+ points the x-prelerped window kernel -- a table of the trilinear sample's four x-lerps and its
 * y-difference per (r byte, y node, z node), 6.9 MB for 33^3, built on the device when the LUT is first used there --, RGBA64 and
 * 3-point cubes the tile kernel of placement 5; otherwise LDS when the table fits: 3-D size <= 21, 1-D size <= 4096; else
 * the cell-packed copy for 3-D size <= 65; else the node layout in global/L2), 1 = node layout in global/L2,
 * 2 = LDS (MVFX_ERR_INVALID_ARGUMENT if it does not fit), 3 = cell-packed global copy,
 * 4 = the literal-transcription kernels (also used automatically when the LUT's domain
 * scale/offset are not finite), 5 = the tile kernel: a wave owns a compact block of pixels and keeps the 3x3x3 LUT cells
 * around the colour of the block's centre pixel in wave-private LDS (MVFX_ERR_INVALID_ARGUMENT when the frame or cube does
 * not allow it), 6 = RGBA8 frames only (other formats take the automatic choice): the LUT baked into a table of all 2^24
 * colours (64 MiB of device memory per LUT, built on the first call by running the interpolating kernels over a frame that
 * holds every colour -- ~5 ms, the host waits once) and applied with one 4-byte gather per pixel; needs 16-byte aligned rows
 * and a width that is a multiple of 4 (MVFX_ERR_INVALID_ARGUMENT otherwise).  Faster than the tile kernel on flat content
 * (bars, animation: 0.69 against 0.47 of the HBM peak), slower on noisy content (0.29 against 0.51) -- never chosen
 * automatically (profiles/r3/colorlut_baked_table.txt). */

/* Writes the LUT as Adobe .cube text: LUT_1D_SIZE / LUT_3D_SIZE, DOMAIN_MIN / DOMAIN_MAX when they are not
 * 0 / 1, then the rows, floats with 9 significant digits -- mvfx_cube_lut_parse of the text yields the same LUT bit for
 * bit (parser.rs:110-282 accepts exactly this grammar).  *text_out is malloc'ed (NUL-terminated); release with
 * mvfx_free_text. */
int mvfx_cube_lut_write(const mvfx_cube_lut *lut, char **text_out, size_t *len_out);
void mvfx_free_text(char *text);

/* ---- imagersoverlay blending : video/image/src/overlay/imp.rs:703-727 ----
 * The element's per-frame work when downstream does not take the overlay-composition meta: `composition.blend(frame)` ==
 * libgstvideo's gst_video_overlay_composition_blend.  One unscaled BGRA rectangle (non-premultiplied: what load_image
 * builds, imp.rs:241-283) at (x, y) -- any position, clipped against the frame -- with the rectangle's global alpha
 * (`alpha` property, imp.rs:183-185) onto a packed RGB frame, in place.  Arithmetic = libgstvideo 1.14.0's (pinned with
 * vectors made by the image's own library, tests/golden/make_overlay_blend_golden.py):
 *   a_s = overlay alpha (x (int)(global_alpha * 255) / 255 when global_alpha != 1); a_s == 0 leaves the pixel untouched;
 *   a_o = a_s + a_d (255 - a_s) / 255;  c_o = (c_s a_s + c_d a_d (255 - a_s) / 255) / max(a_o, 1), truncating divisions;
 *   a_d = 255 for RGB / BGR; the x byte of RGBx / BGRx / xRGB / xBGR is treated as alpha, as 1.14.0 does.
 * Destination formats: the ten packed RGB formats of mvfx_format.  `overlay` must be MVFX_FORMAT_BGRA. */
int mvfx_overlay_blend(const mvfx_frame *frame, const mvfx_frame *overlay, int32_t x, int32_t y, float global_alpha,
                       mvfx_stream stream);
/* host memory for both (a mapped GstBuffer and the decoded image): only the clipped rectangle crosses PCIe */
int mvfx_overlay_blend_host(const mvfx_frame *frame, const mvfx_frame *overlay, int32_t x, int32_t y, float global_alpha);

/* ---- colordetect : video/videofx/src/colordetect/imp.rs ----
 * The reference calls color_thief::get_palette(plane, format, quality, max_colors) (crate
 * color-thief 0.2.2, imp.rs:68-74) and names palette[0] with color_name::css::Color::similar
 * (imp.rs:77-79).  Here the O(pixels) half (5-5-5 histogram of every `quality`-th pixel of the flat
 * plane, row padding included) runs on the GPU; the serial median cut runs on the host. */

/* Histogram + 5-bit channel min/max of samples [first_sample, first_sample+n_samples) of the
 * frame (sample k = pixel k*quality of the flat plane); pass n_samples = UINT64_MAX for all.
 * hist_device: 32768 u32 (16-byte aligned), minmax_device: 6 u32 {rmin,rmax,gmin,gmax,bmin,bmax}; both are
 * overwritten.  A sample range exists so that ranks can split one frame and all-reduce
 * (sum the histogram, min/max the bounds).  Formats RGB RGBA ARGB BGR BGRA; quality 1..=10. */
int mvfx_colordetect_histogram(const mvfx_frame *frame, uint32_t quality, uint64_t first_sample,
                               uint64_t n_samples, uint32_t *hist_device,
                               uint32_t *minmax_device, mvfx_stream stream);
/* The same for one frame of each of n_frames streams (same size, stride and format) in ONE pair of launches (blockIdx.y =
 * stream): hist_device holds n_frames records of MVFX_COLORDETECT_RECORD_WORDS u32 -- 32768 bins, the six bounds
 * {rmin,rmax,gmin,gmax,bmin,bmax}, two words of padding -- record f for frames[f].  A single 4K frame at quality 10 is ~10 us of
 * GPU work of which half is the fixed cost of two launches; 16 streams per launch run at the HBM rate of reading the frames. */
#define MVFX_COLORDETECT_RECORD_WORDS (32768u + 8u)
int mvfx_colordetect_histogram_frames(const mvfx_frame *frames, uint32_t n_frames, uint32_t quality,
                                      uint32_t *hist_device, mvfx_stream stream);
/* Host-only median cut over a (possibly all-reduced) histogram.  palette_out: max_colors
 * entries packed 0x00RRGGBB, most dominant first (colordetect/imp.rs:95-99). */
int mvfx_mmcq_palette_from_histogram(const uint32_t *hist_host, const uint32_t minmax[6],
                                     uint32_t max_colors, uint32_t *palette_out, uint32_t *n_out);
/* histogram + D2H + median cut in one call (synchronous) */
int mvfx_colordetect_palette(const mvfx_frame *frame, uint32_t quality, uint32_t max_colors,
                             uint32_t *palette_out, uint32_t *n_out, mvfx_stream stream);
int mvfx_colordetect_palette_host(const mvfx_frame *frame, uint32_t quality, uint32_t max_colors,
                                  uint32_t *palette_out, uint32_t *n_out);
/* lower-case CSS colour name nearest to (r,g,b); static storage */
const char *mvfx_css_color_similar(uint8_t r, uint8_t g, uint8_t b);

/* ---- videocompare : video/videofx/src/videocompare/{imp,hashed_image}.rs ----
 * Default hash-algo Blockhash (image_hasher 3.1.1, 8x8 bits, hashed_image.rs:37-45,104):
 * 64 block sums of r+g+b (765 when alpha == 0) over an 8x8 grid, bits against the band
 * median, distance = Hamming distance as f64 (hashed_image.rs:70); row padding never counts (the
 * reference packs the frame first, hashed_image.rs:110-130).  Any RGB / RGBA size is accepted, as by the
 * reference (videocompare/imp.rs:158-163):
 *   width and height multiples of 8 (every BASELINE shape): the crate's integer path, u32 sums;
 *   any other size: the crate's f32 path -- every block sum is ONE chain of f32 additions in raster order
 *   (64 independent ordered chains, one wave per block on the device); the 64 words are then f32 BIT
 *   PATTERNS, mvfx_blockhash_bits compares them with the crate's 0.001 margin.  That path cannot be split
 *   into row bands (MVFX_ERR_INVALID_ARGUMENT for a partial row range). */

/* Partial block sums of image rows [row_begin,row_end) -> sums_device[64] (overwritten).
 * Ranks that each own a row band all-reduce(sum) the 64 u32 and then call
 * mvfx_blockhash_bits on the total (sizes that are multiples of 8 only). */
int mvfx_blockhash_sums(const mvfx_frame *frame, uint32_t row_begin, uint32_t row_end,
                        uint32_t *sums_device, mvfx_stream stream);
/* Same for a rank that holds ONLY its row band in memory: `band` describes rows
 * [band_first_row, band_first_row + band->height) of a frame of `full_height` rows. */
int mvfx_blockhash_sums_band(const mvfx_frame *band, uint32_t full_height,
                             uint32_t band_first_row, uint32_t *sums_device,
                             mvfx_stream stream);
/* The same for n_pads frames (or bands) of one size and format in ONE launch (what
 * aggregate_frames hashes per output buffer: the reference pad + every other pad,
 * videocompare/imp.rs:316,349-353) -> sums_device[n_pads][64].  Whole frames: full_height =
 * height, band_first_row = 0. */
int mvfx_blockhash_sums_pads(const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height,
                             uint32_t band_first_row, uint32_t *sums_device, mvfx_stream stream);
int mvfx_blockhash_bits(const uint32_t sums_host[64], uint32_t width, uint32_t height,
                        uint64_t *hash_out);
uint32_t mvfx_hash_distance(uint64_t a, uint64_t b);
int mvfx_blockhash(const mvfx_frame *frame, uint64_t *hash_out, mvfx_stream stream); /* sync */
int mvfx_blockhash_host(const mvfx_frame *frame, uint64_t *hash_out);
/* HasherEngine::hash_image x2 + compare (videocompare/imp.rs:316,349-353); sizes must match
 * (MVFX_ERR_NOT_NEGOTIATED, imp.rs:337-346). */
int mvfx_videocompare_distance(const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                               double *distance_out, mvfx_stream stream);

/* ---- the collective of the path, inside the library (videocompare/imp.rs:259-389 on row-distributed frames) ----
 * RCCL is loaded at run time (dlopen "librccl.so.1", MVFX_RCCL_LIBRARY overrides; MVFX_ERR_IO if absent): only multi-GPU users
 * need it.  One process per GPU: rank 0 makes a 128-byte id (mvfx_comm_unique_id), the host layer hands it to the other ranks by
 * whatever it has, every rank calls mvfx_comm_create on its own device (ncclCommInitRank).  A NULL communicator means one GPU. */
typedef struct mvfx_comm mvfx_comm;
#define MVFX_COMM_ID_BYTES 128
int mvfx_comm_unique_id(uint8_t id_out[MVFX_COMM_ID_BYTES]);
int mvfx_comm_create(const uint8_t id[MVFX_COMM_ID_BYTES], int rank, int world, mvfx_comm **out);
int mvfx_comm_destroy(mvfx_comm *comm);
int mvfx_comm_rank(const mvfx_comm *comm);
int mvfx_comm_world(const mvfx_comm *comm);
/* In-place all-reduce of `count` elements in device memory on `stream` (asynchronous): the SSIM partial sums (10 f64) and the
 * colordetect histogram (32768 u32 + bounds) of a distributed frame go through the same communicator. */
#define MVFX_DTYPE_U32 0
#define MVFX_DTYPE_U64 1
#define MVFX_DTYPE_F64 2
#define MVFX_REDUCE_SUM 0
#define MVFX_REDUCE_MIN 1
#define MVFX_REDUCE_MAX 2
int mvfx_comm_allreduce(mvfx_comm *comm, void *buffer_device, size_t count, int32_t dtype, int32_t op, mvfx_stream stream);
/* One aggregate of videocompare (hash-algo=blockhash) over pads whose frames are distributed by rows: `bands` = this rank's rows
 * [band_first_row, band_first_row + bands[p].height) of every pad's frame of `full_height` rows (pad 0 = the reference pad,
 * imp.rs:210-233).  Band kernel -> ncclAllReduce(sum) of n_pads x 64 u32 on `stream` -> hash bits and Hamming distances on the
 * device -> one D2H of n_pads - 1 words; every rank returns the same distances_out[n_pads - 1] (distance of pad p + 1 to the
 * reference, hashed_image.rs:70) and, when hashes_out != NULL, the n_pads 64-bit hashes.  Sizes must be multiples of 8 (the
 * crate's f32 path for other sizes cannot be split by rows); n_pads <= 16.  Synchronous. */
int mvfx_videocompare_sharded_distances(mvfx_comm *comm, const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height,
                                        uint32_t band_first_row, double *distances_out, uint64_t *hashes_out, mvfx_stream stream);

/* The same aggregate for hash-algo=dssim (BASELINE config 5: "videocompare SSIM ... tile-sharded across 8 GPUs with RCCL all-reduce";
 * hashed_image.rs:49-59,72-75): every rank holds both WHOLE frames (a band's five-level pyramid needs a halo of its neighbours' rows)
 * and maps rows [row_begin, row_end) only -- band boundaries multiples of 16 or the frame height, the bands of all ranks a partition
 * of the rows.  mvfx_ssim_partial_sums on the band -> ncclAllReduce(sum) of 10 f64 (per-scale sums and pixel counts) ->
 * mvfx_ssim_partial_deviation against the global means -> ncclAllReduce(sum) of 5 f64 -> mvfx_ssim_combine: every rank returns the
 * same distance.  comm == NULL: one GPU (then identical to mvfx_ssim_distance when the band is the whole frame).  Synchronous. */
int mvfx_videocompare_sharded_dssim(mvfx_comm *comm, const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                                    uint32_t row_begin, uint32_t row_end, double *distance_out, mvfx_stream stream);

/* GstVideoCompareHashAlgorithm values (videocompare/mod.rs:57-92) */
typedef enum mvfx_hash_algo {
    MVFX_HASH_MEAN = 0,
    MVFX_HASH_GRADIENT = 1,
    MVFX_HASH_VERTGRADIENT = 2,
    MVFX_HASH_DOUBLEGRADIENT = 3,
    MVFX_HASH_BLOCKHASH = 4,
    MVFX_HASH_DSSIM = 5
} mvfx_hash_algo;

/* hash-algo = mean / gradient / vertgradient / doublegradient (HasherEngine::from, hashed_image.rs:89-107 ->
 * image_hasher 3.1.1 on image 0.25.10; crates not vendored under the reference: PARITY UNPINNED): integer
 * Rec.709 grayscale, Lanczos3 resize to 8x8 / 9x8 / 8x9 / 5x5 with image 0.25's f32 accumulation order, then
 * compare-to-mean / neighbour compares.  hash_out bit k = k-th bool of the crate's iterator; n_bits_out (may
 * be NULL) = 64, 64, 64, 40.  RGB / RGBA only; synchronous. */
int mvfx_image_hash(const mvfx_frame *frame, int32_t hash_algo, uint64_t *hash_out,
                    uint32_t *n_bits_out, mvfx_stream stream);
int mvfx_image_hash_host(const mvfx_frame *frame, int32_t hash_algo, uint64_t *hash_out,
                         uint32_t *n_bits_out);
/* The resampling step alone (image::imageops::grayscale + resize(.., FilterType::Lanczos3)):
 * new_width x new_height bytes into host memory; targets up to 64 x 64.  Synchronous. */
int mvfx_image_gray_resize_lanczos3(const mvfx_frame *frame, uint32_t new_width,
                                    uint32_t new_height, uint8_t *out_host, mvfx_stream stream);
/* hash_image x2 + compare for ANY hash-algo value (dispatches to the blockhash / dssim paths for 4 / 5) */
int mvfx_videocompare_distance_algo(const mvfx_frame *reference_frame,
                                    const mvfx_frame *other_frame, int32_t hash_algo,
                                    double *distance_out, mvfx_stream stream);

/* `hash-algo=dssim` (HashAlg::Dssim, videocompare/hashed_image.rs:49-59,72-75, cargo feature
 * `dssim`, NOT in the default build).  dssim-core 3.4.0 is not vendored under the reference:
 * this is the published multi-scale SSIM structure, PARITY UNPINNED against the
 * crate; identical frames give exactly 0.0 (tests/videocompare.rs:141-182).  RGB / RGBA only.
 *
 * Arithmetic: f32 per pixel (dssim-core is an f32 library) with f64 reductions; a per-tile centring constant keeps the f32
 * variances free of cancellation; measured against the f64 checker (profiles/r3/ssim32_error_vs_f64_oracle.txt): ~1e-10 absolute,
 * up to ~7e-5 relative for distances around 1e-6 (near-identical frames), ~1e-6 relative for ordinary pairs.
 * MVFX_OPT_SSIM_F64 selects f64 throughout.
 * A pass 1 (mvfx_ssim_partial_sums) decides which pipeline its pass 2 runs on; a pass 1 that failed or was abandoned is forgotten
 * by the next pass 1 of either pipeline.
 *
 * Two-pass, shardable by row bands (boundaries multiples of 16, or the frame height):
 *   1. mvfx_ssim_partial_sums: per-scale sum of the SSIM map over the band + pixel counts;
 *      the maps stay in device scratch owned by the calling thread.
 *   2. (all-reduce sums and counts; mean = sum / count)
 *   3. mvfx_ssim_partial_deviation: per-scale sum of |map - mean| over the same band.
 *   4. (all-reduce; mvfx_ssim_combine(mean, deviation_sum / count, n_scales) -> distance)
 * mvfx_ssim_distance does 1-4 for one device. */
int mvfx_ssim_partial_sums(const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                           uint32_t row_begin, uint32_t row_end, double sums_out[5],
                           double counts_out[5], uint32_t *n_scales_out, mvfx_stream stream);
int mvfx_ssim_partial_deviation(const double mean[5], double deviation_sums_out[5],
                                mvfx_stream stream);
double mvfx_ssim_combine(const double mean[5], const double mean_abs_deviation[5],
                         uint32_t n_scales);
int mvfx_ssim_distance(const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                       double *distance_out, mvfx_stream stream);
int mvfx_ssim_distance_host(const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                            double *distance_out);

/* ---- roundedcorners : video/videofx/src/border/imp.rs ----
 * Planar view for I420 / A420 (GstVideoFrame plane pointers + strides). */
typedef struct mvfx_planar_frame {
    void *data[4];
    uint32_t stride[4];
    uint32_t width, height;
    int32_t format; /* MVFX_FORMAT_I420, MVFX_FORMAT_A420 or MVFX_FORMAT_NV12 */
} mvfx_planar_frame;

/* Replaces generate_alpha_mask + draw_rounded_corners (border/imp.rs:57-180): writes the A8
 * plane (stride x round_up_2(height) bytes): 0xFF everywhere when border_radius_px == 0
 * (border/imp.rs:123-128), else the rounded rectangle drawn by libcairo with the reference's exact
 * call sequence (new_sub_path, four arcs, close_path, fill_preserve, 1 px stroke) -- the same C
 * library the reference calls through cairo-rs, loaded at run time (dlopen "libcairo.so.2";
 * MVFX_CAIRO_LIBRARY overrides the path; MVFX_ERR_IO if absent), so the bytes equal the
 * reference's for every radius, including 2*radius > min(width, height).  Runs once per caps /
 * radius change (border/imp.rs:491-519); the per-frame device work is the compose below.
 * The device flavour renders on the host and uploads (synchronous on `stream`). */
int mvfx_roundedcorners_mask(uint8_t *mask_device, uint32_t width, uint32_t height,
                             uint32_t stride, uint32_t border_radius_px, mvfx_stream stream);
/* same, into host memory (the shared alpha GstMemory of the element); synchronous, touches no device */
int mvfx_roundedcorners_mask_host(uint8_t *mask_host, uint32_t width, uint32_t height,
                                  uint32_t stride, uint32_t border_radius_px);
/* cairo_version_string() of the libcairo in use, NULL when none can be loaded */
const char *mvfx_roundedcorners_cairo_version(void);
/* I420 -> A420 into one device buffer: copies Y, U, V and the mask as plane 3 (what
 * prepare_output_buffer does by appending the shared alpha GstMemory, border/imp.rs:482-559). */
int mvfx_roundedcorners_compose_a420(const mvfx_planar_frame *i420_in, const uint8_t *mask_device,
                                     uint32_t mask_stride, const mvfx_planar_frame *a420_out,
                                     mvfx_stream stream);

/* ---- videoconvert-equivalent I420 <-> RGBA ----
 * What GStreamer's `videoconvert` does on either side of the filters in the reference's own example pipeline
 * (`... ! videoconvert ! colorlut ! videoconvert ! ...`, video/colorlut/src/colorlut/imp.rs:17-19), for device-resident
 * pipelines.  Bit-exact with GStreamer 1.14.0's default-caps conversion (I420 -> RGBA: the orc fast path with chroma
 * duplication; RGBA -> I420: 8-bit matrix, chroma averaged vertically then horizontally, co-sited 1-2-1 for HD / UHD);
 * gst-plugins-base is not under the reference tree: pinned against the real element through goldens
 * (tests/golden/make_videoconvert_golden.py).
 * yuv_standard: 0 = GStreamer 1.14's default for the frame height (<= 576 lines BT.601 + chroma-site none, < 2160
 * BT.709 + h-cosited, else BT.2020 + h-cosited), 1 / 2 / 3 force BT.601 / BT.709 / BT.2020.
 * RGBA -> I420 takes any size: an odd-sized frame is converted as the next even size with its last column / row replicated,
 * which is what the element does (chroma planes of RU2(w)/2 x RU2(h)/2 samples).  Device pointers; asynchronous on `stream`. */
int mvfx_convert_i420_to_rgba(const mvfx_planar_frame *i420_in, const mvfx_frame *rgba_out,
                              int32_t yuv_standard, mvfx_stream stream);
int mvfx_convert_rgba_to_i420(const mvfx_frame *rgba_in, const mvfx_planar_frame *i420_out,
                              int32_t yuv_standard, mvfx_stream stream);
/* RGBA -> NV12 (what encoders and display engines take): the same Y, U and V as RGBA -> I420, data[0] = Y plane, data[1] = plane
 * of interleaved (U, V) pairs, stride[1] >= 2 * RU2(w)/2; any size. */
int mvfx_convert_rgba_to_nv12(const mvfx_frame *rgba_in, const mvfx_planar_frame *nv12_out, int32_t yuv_standard,
                              mvfx_stream stream);
/* NV12 -> RGBA (what hardware decoders emit): GStreamer 1.14.0 has no fast path for it -- unlike I420 -> RGBA, which duplicates the
 * chroma -- and interpolates the chroma, horizontally then vertically (video-chroma.c; chroma-site none <= 576 lines, h-cosited
 * above), before the same matrix; reproduced bit for bit (goldens from the element, incl. odd sizes and 3840x2160).  Any size. */
int mvfx_convert_nv12_to_rgba(const mvfx_planar_frame *nv12_in, const mvfx_frame *rgba_out, int32_t yuv_standard,
                              mvfx_stream stream);
/* n frame pairs sharing geometry and strides (one frame from each of n streams) in ONE launch */
int mvfx_convert_i420_to_rgba_frames(const mvfx_planar_frame *i420_in, const mvfx_frame *rgba_out,
                                     uint32_t n_frames, int32_t yuv_standard, mvfx_stream stream);
int mvfx_convert_rgba_to_i420_frames(const mvfx_frame *rgba_in, const mvfx_planar_frame *i420_out,
                                     uint32_t n_frames, int32_t yuv_standard, mvfx_stream stream);
/* `videoconvert ! colorlut ! videoconvert` on a device-resident I420 frame (colorlut/imp.rs:17-19): I420 -> RGBA,
 * the LUT (colorlut/imp.rs:226-294), RGBA -> I420 -- fused into ONE kernel when the frame allows it (width % 8 == 0,
 * even height, planes aligned to 8 / 4 bytes, finite LUT domain), the same three steps through scratch frames
 * otherwise; identical bytes either way (= the three separate entry points in sequence). */
int mvfx_colorlut_transform_i420(mvfx_cube_lut *lut, const mvfx_planar_frame *i420_in,
                                 const mvfx_planar_frame *i420_out, int32_t yuv_standard,
                                 mvfx_stream stream);
/* `videoconvert ! hsvfilter ! videoconvert` on a device-resident I420 frame, same construction (hsvfilter takes RGB
 * formats only, hsvfilter/imp.rs:278-289).  Input and output planes must not alias. */
int mvfx_hsvfilter_transform_i420(const mvfx_planar_frame *i420_in, const mvfx_planar_frame *i420_out,
                                  const mvfx_hsvfilter_settings *settings, int32_t yuv_standard,
                                  mvfx_stream stream);
/* `videoconvert ! hsvdetector` on a device-resident I420 frame: I420 -> RGB in registers -> hsv_detect
 * (hsvdetector/imp.rs:100-160) -> RGBA / ARGB / BGRA / ABGR.  Any frame size. */
int mvfx_hsvdetector_transform_i420(const mvfx_planar_frame *i420_in, const mvfx_frame *out_frame,
                                    const mvfx_hsvdetector_settings *settings, int32_t yuv_standard,
                                    mvfx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355VFX_H */
