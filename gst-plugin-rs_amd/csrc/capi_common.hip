// Library-level pieces of the C ABI: error reporting, device selection, device buffers,
// the staging scratch behind the *_host entry points.
#include "mvfx_internal.h"
#include "direct_dispatch.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace mvfx {

namespace {
thread_local char t_last_error[512] = "";
thread_local uint32_t t_options = 0; // mvfx_thread_set_options
thread_local hipEvent_t t_completion = nullptr; // mvfx_thread_set_completion_event
thread_local uint32_t t_completion_uses = 0;

struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};
constexpr int kScratchSlots = 6; // 0-3: staging of the *_host entry points and small results (4, 5: free)
constexpr int kStreamScratch = 8; // scratch blocks keyed by the stream the work is enqueued on (stream_scratch)

// Per-thread, per-device staging state of the *_host entry points (a thread that switches devices with
// mvfx_set_device gets a separate stream and scratch set for each ordinal); released when the thread (e.g. a
// GStreamer streaming thread) exits so pipelines that come and go do not leak device memory.
struct StreamScratch {
    hipStream_t stream = nullptr;
    Scratch block;
    uint64_t last_use = 0;
    bool used = false;
};
struct DeviceState {
    Scratch scratch[kScratchSlots];
    StreamScratch by_stream[kStreamScratch];
    uint64_t tick = 0;
    hipStream_t extra[3] = {nullptr, nullptr, nullptr}; // mvfx_thread_stream_n(1..3)
    bool extra_tried[3] = {false, false, false};
    hipStream_t stream = nullptr;
    bool stream_tried = false;
    bool bundle_tried = false; // the pool of exited threads' streams has been asked once
    int device = 0;
};
// The private streams of threads that have exited, per device, for the next thread that asks.  Streams are never destroyed:
// sixteen GStreamer streaming threads leaving at end-of-stream called hipStreamDestroy concurrently and the runtime (ROCm 7.2) crashed
// inside it about two runs in three with two streams per thread (tools/exp_rot2_crash.py, backtrace via tools/segv_trace.c: two threads
// in ~ThreadState -> hipStreamDestroy at once).  A recycled stream may still have its previous owner's work queued: that only orders
// ahead of the new owner's.
// A thread's streams are pooled and handed out as ONE BUNDLE: the runtime binds a stream to one of its few hardware queues when the
// stream is created (the least loaded one: a thread that creates its streams one after the other gets them on different queues), and
// kernels of two streams that share a hardware queue do not overlap.  Handing out single streams in any order gave a thread two
// streams of one queue and the gain of alternating between them was gone (one thread, 4K hsvfilter: 64.9 k fps on such a pair, 80.9 k
// on a pair created together -- bench.py config.one_video_stream_launch_models).  The pool is bounded by the largest number of threads
// alive at the same time; it is leaked on purpose (threads may exit after the static destructors ran).
struct StreamBundle {
    hipStream_t stream = nullptr;                         // mvfx_thread_stream() = mvfx_thread_stream_n(0)
    hipStream_t extra[3] = {nullptr, nullptr, nullptr};   // mvfx_thread_stream_n(1..3)
    bool any() const { return stream || extra[0] || extra[1] || extra[2]; }
};
struct IdleStreams {
    std::mutex lock;
    std::map<int, std::vector<StreamBundle>> by_device;
};
IdleStreams &idle_streams()
{
    static IdleStreams *pool = new IdleStreams;
    return *pool;
}
hipStream_t new_stream()
{
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        s = nullptr; // the null stream
    }
    return s;
}
// the bundle of an exited thread (the one with the most streams in it first), or an empty one
void adopt_bundle(DeviceState &d)
{
    if (d.bundle_tried) return;
    d.bundle_tried = true;
    IdleStreams &pool = idle_streams();
    std::lock_guard<std::mutex> g(pool.lock);
    std::vector<StreamBundle> &v = pool.by_device[d.device];
    if (v.empty()) return;
    size_t best = 0;
    auto count = [](const StreamBundle &b) { return (b.stream != nullptr) + (b.extra[0] != nullptr) + (b.extra[1] != nullptr) + (b.extra[2] != nullptr); };
    for (size_t i = 1; i < v.size(); i++)
        if (count(v[i]) > count(v[best])) best = i;
    const StreamBundle b = v[best];
    v.erase(v.begin() + (long)best);
    d.stream = b.stream;
    d.stream_tried = b.stream != nullptr;
    for (int k = 0; k < 3; k++) {
        d.extra[k] = b.extra[k];
        d.extra_tried[k] = b.extra[k] != nullptr;
    }
}
void give_back_bundle(int device, const DeviceState &d)
{
    StreamBundle b;
    b.stream = d.stream;
    for (int k = 0; k < 3; k++) b.extra[k] = d.extra[k];
    if (!b.any()) return;
    IdleStreams &pool = idle_streams();
    std::lock_guard<std::mutex> g(pool.lock);
    pool.by_device[device].push_back(b);
}

struct ThreadState {
    std::map<int, DeviceState> per_device;
    DeviceState &current()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        DeviceState &d = per_device[dev];
        d.device = dev;
        return d;
    }
    ~ThreadState()
    {
        for (auto &kv : per_device) {
            for (Scratch &s : kv.second.scratch)
                if (s.ptr) (void)hipFree(s.ptr);
            for (StreamScratch &s : kv.second.by_stream)
                if (s.block.ptr) (void)hipFree(s.block.ptr);
            give_back_bundle(kv.first, kv.second);
        }
    }
};
thread_local ThreadState t_state;
} // namespace

int fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_last_error, sizeof(t_last_error), fmt, ap);
    va_end(ap);
    return status;
}

uint32_t thread_options() { return t_options; }

hipEvent_t completion_event() { return t_completion; }
void note_completion_event_used() { t_completion_uses++; }

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MVFX_ERR_NO_DEVICE,
                    "no HIP device available (%s); libmi355vfx has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    return MVFX_OK;
}

int check_packed_frame(const mvfx_frame *f, const char *what)
{
    if (!f)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL frame", what);
    const int bpp = bytes_per_pixel(f->format);
    if (bpp == 0)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "%s: format %d is not a packed format", what, f->format);
    if (f->width != 0 && f->height != 0) {
        if (!f->data)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: NULL plane pointer", what);
        if (f->stride == 0)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: zero stride", what);
        if ((uint64_t)f->width * (uint64_t)bpp > f->stride)
            return fail(MVFX_ERR_INVALID_ARGUMENT, "%s: row of %u pixels x %d bytes exceeds stride %u", what,
                        f->width, bpp, f->stride);
    }
    return MVFX_OK;
}

int host_scratch(size_t bytes, int slot, void **out)
{
    if (slot < 0 || slot >= kScratchSlots)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "bad scratch slot %d", slot);
    Scratch &s = t_state.current().scratch[slot];
    if (s.cap < bytes) {
        if (s.ptr) {
            MVFX_HIP_TRY(hipFree(s.ptr));
            s.ptr = nullptr;
            s.cap = 0;
        }
        const size_t want = bytes + (bytes >> 2); // head-room for the next, slightly larger, frame
        hipError_t e = hipMalloc(&s.ptr, want);
        if (e != hipSuccess)
            return fail(MVFX_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        s.cap = want;
    }
    *out = s.ptr;
    return MVFX_OK;
}

// Scratch for kernels that hand intermediate results from one launch to the next on `stream` (colordetect's partial
// histograms): one block per (thread, device, stream), so two asynchronous calls of one thread on different streams never
// share a block; a block only grows (hipFree synchronises the device, so work in flight on the old block is over before it goes).
static int stream_scratch_slot(hipStream_t stream, StreamScratch **out)
{
    DeviceState &d = t_state.current();
    StreamScratch *slot = nullptr, *free_slot = nullptr, *lru = nullptr;
    for (StreamScratch &s : d.by_stream) {
        if (s.used && s.stream == stream) { slot = &s; break; }
        if (!s.used) { if (!free_slot) free_slot = &s; }
        else if (!lru || s.last_use < lru->last_use) lru = &s;
    }
    if (!slot) {
        slot = free_slot ? free_slot : lru;
        if (!free_slot) {
            // the least recently used stream's block changes owner: work that stream still has in flight may be reading or writing
            // it (colordetect's launches are asynchronous), so it is given back -- hipFree waits for the device -- and the new
            // owner allocates its own (a thread that works on more than kStreamScratch streams pays this; none of the elements does)
            if (slot->block.ptr) MVFX_HIP_TRY(hipFree(slot->block.ptr));
            slot->block.ptr = nullptr;
            slot->block.cap = 0;
        }
        slot->stream = stream;
        slot->used = true;
    }
    slot->last_use = ++d.tick;
    *out = slot;
    return MVFX_OK;
}

int stream_scratch(hipStream_t stream, size_t bytes, void **out)
{
    StreamScratch *slot = nullptr;
    if (int rc = stream_scratch_slot(stream, &slot); rc != MVFX_OK) return rc;
    Scratch &s = slot->block;
    if (s.cap < bytes) {
        if (s.ptr) {
            MVFX_HIP_TRY(hipFree(s.ptr));
            s.ptr = nullptr;
            s.cap = 0;
        }
        const size_t want = bytes + (bytes >> 2);
        hipError_t e = hipMalloc(&s.ptr, want);
        if (e != hipSuccess)
            return fail(MVFX_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        s.cap = want;
    }
    *out = s.ptr;
    return MVFX_OK;
}

hipStream_t host_stream()
{
    DeviceState &d = t_state.current();
    adopt_bundle(d);
    if (!d.stream_tried) {
        d.stream_tried = true;
        d.stream = new_stream(); // nullptr (the null stream) when none can be created
        if (!d.extra_tried[0]) { // its partner right behind it (a different hardware queue: see StreamBundle), whether or not it is used
            d.extra_tried[0] = true;
            d.extra[0] = new_stream();
        }
    }
    return d.stream;
}

// 0..3 when `stream` is one of the calling thread's private streams on the current device, else -1
int thread_stream_index(hipStream_t stream)
{
    DeviceState &d = t_state.current();
    if (stream && stream == d.stream) return 0;
    for (int k = 0; k < 3; k++)
        if (stream && stream == d.extra[k]) return k + 1;
    return -1;
}

hipStream_t host_stream_n(uint32_t index)
{
    index %= 4; // documented: the index is taken modulo 4 (4 is stream 0 again)
    if (index == 0) return host_stream();
    DeviceState &d = t_state.current();
    const uint32_t k = index - 1;
    (void)host_stream(); // adopts a pooled bundle or creates stream 0 and its partner first
    if (!d.extra_tried[k]) {
        d.extra_tried[k] = true;
        d.extra[k] = new_stream();
    }
    return d.extra[k] ? d.extra[k] : host_stream();
}

} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_abi_version(void) { return MVFX_ABI_VERSION; }

const char *mvfx_last_error(void) { return t_last_error; }

const char *mvfx_status_string(int status)
{
    switch (status) {
    case MVFX_OK: return "ok";
    case MVFX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MVFX_ERR_UNSUPPORTED_FORMAT: return "unsupported format";
    case MVFX_ERR_NOT_NEGOTIATED: return "not negotiated";
    case MVFX_ERR_DEVICE: return "device error";
    case MVFX_ERR_NO_DEVICE: return "no device";
    case MVFX_ERR_PARSE: return "parse error";
    case MVFX_ERR_IO: return "i/o error";
    case MVFX_ERR_REFERENCE_PANIC: return "reference would panic";
    case MVFX_ERR_NO_LUT: return "no LUT configured";
    case MVFX_ERR_OUT_OF_MEMORY: return "out of memory";
    default: return "unknown status";
    }
}

int mvfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int mvfx_set_device(int ordinal)
{
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipSetDevice(ordinal));
    return MVFX_OK;
}

int mvfx_current_device(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return dev;
}

int mvfx_stream_synchronize(mvfx_stream stream)
{
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return MVFX_OK;
}

int mvfx_device_alloc(void **out_ptr, size_t bytes)
{
    if (!out_ptr)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "device_alloc: NULL out pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    *out_ptr = nullptr;
    hipError_t e = hipMalloc(out_ptr, bytes ? bytes : 1);
    if (e != hipSuccess)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MVFX_OK;
}

int mvfx_device_free(void *ptr)
{
    if (!ptr) return MVFX_OK;
    MVFX_HIP_TRY(hipFree(ptr));
    return MVFX_OK;
}

int mvfx_copy_to_device(void *dst_device, const void *src_host, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_device || !src_host)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_to_device: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    MVFX_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return MVFX_OK;
}

int mvfx_copy_to_host(void *dst_host, const void *src_device, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_host || !src_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_to_host: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    MVFX_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return MVFX_OK;
}

int mvfx_copy_device_to_device(void *dst_device, const void *src_device, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_device || !src_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_device_to_device: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_device, src_device, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    MVFX_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return MVFX_OK;
}

mvfx_stream mvfx_thread_stream(void) { return reinterpret_cast<mvfx_stream>(host_stream()); }
mvfx_stream mvfx_thread_stream_n(uint32_t index) { return reinterpret_cast<mvfx_stream>(host_stream_n(index)); }

int mvfx_event_create(mvfx_event *out)
{
    if (!out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "event_create: NULL out pointer");
    *out = nullptr;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    hipEvent_t e = nullptr;
    MVFX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); // ordering only: the cheap kind
    *out = reinterpret_cast<mvfx_event>(e);
    return MVFX_OK;
}

int mvfx_event_destroy(mvfx_event event)
{
    if (!event) return MVFX_OK;
    direct_event_destroy(reinterpret_cast<hipEvent_t>(event)); // (waits for a lane dispatch the event still stands for)
    MVFX_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MVFX_OK;
}

int mvfx_event_record(mvfx_event event, mvfx_stream stream)
{
    if (!event)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "event_record: NULL event");
    direct_event_forget(reinterpret_cast<hipEvent_t>(event)); // from here on the event means this record
    MVFX_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return MVFX_OK;
}

int mvfx_stream_wait_event(mvfx_stream stream, mvfx_event event)
{
    if (!event)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "stream_wait_event: NULL event");
    // A direct fence (the completion signal of a frame that went out on the library's own queue, MVFX_OPT_DIRECT_DISPATCH) is nothing a HIP stream
    // can wait for on the device: finished (the usual case: a recycled block) -> nothing to do, not even a HIP call; still running -> the CALLING
    // THREAD waits (tens of microseconds), after which the stream needs no wait either.
    if (const int st = direct_event_state(reinterpret_cast<hipEvent_t>(event)); st != 0) {
        if (st == 2) direct_event_wait(reinterpret_cast<hipEvent_t>(event));
        return MVFX_OK;
    }
    MVFX_HIP_TRY(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return MVFX_OK;
}

int mvfx_event_synchronize(mvfx_event event)
{
    if (!event)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "event_synchronize: NULL event");
    if (direct_event_state(reinterpret_cast<hipEvent_t>(event)) != 0) return direct_event_wait(reinterpret_cast<hipEvent_t>(event));
    MVFX_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return MVFX_OK;
}

int mvfx_thread_set_completion_event(mvfx_event event)
{
    // whatever carries the event next (a kernel's stop event, or the lane's completion signal) gives it its meaning: an earlier lane dispatch it
    // may still stand for is finished first (never pending in practice: the element layer re-uses a fence only when no block points at it)
    if (event) direct_event_forget(reinterpret_cast<hipEvent_t>(event));
    t_completion = reinterpret_cast<hipEvent_t>(event);
    t_completion_uses = 0;
    return MVFX_OK;
}

int mvfx_thread_clear_completion_event(void)
{
    const uint32_t uses = t_completion_uses;
    t_completion = nullptr;
    t_completion_uses = 0;
    return (int)uses;
}

int mvfx_event_query(mvfx_event event)
{
    if (!event)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "event_query: NULL event");
    if (const int st = direct_event_state(reinterpret_cast<hipEvent_t>(event)); st != 0) return st == 1 ? 1 : 0;
    const hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(event));
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError(); // not an error: do not leave it for the next call's check
        return 0;
    }
    return fail(MVFX_ERR_DEVICE, "event_query: %s", hipGetErrorString(e));
}

int mvfx_event_is_direct(mvfx_event event) { return event && direct_event_state(reinterpret_cast<hipEvent_t>(event)) != 0 ? 1 : 0; }
int mvfx_event_direct_queue(mvfx_event event) { return event ? direct_event_queue(reinterpret_cast<hipEvent_t>(event)) : -1; }
int mvfx_direct_queue_of_stream(mvfx_stream stream) { return direct_queue_hint(as_stream(stream)); }

int mvfx_direct_lane_park(void)
{
    int device = 0;
    MVFX_HIP_TRY(hipGetDevice(&device));
    return direct_park(device);
}

int mvfx_direct_queue_wait_event(int queue, mvfx_event event)
{
    if (!event) return fail(MVFX_ERR_INVALID_ARGUMENT, "direct_queue_wait_event: NULL event");
    return direct_queue_wait(reinterpret_cast<hipEvent_t>(event), queue);
}

int mvfx_host_alloc(void **out_ptr, size_t bytes)
{
    if (!out_ptr)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "host_alloc: NULL out pointer");
    *out_ptr = nullptr;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    hipError_t e = hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(MVFX_ERR_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MVFX_OK;
}

int mvfx_host_free(void *ptr)
{
    if (!ptr) return MVFX_OK;
    MVFX_HIP_TRY(hipHostFree(ptr));
    return MVFX_OK;
}

int mvfx_copy_to_device_async(void *dst_device, const void *src_host, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_device || !src_host)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_to_device_async: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return MVFX_OK;
}

int mvfx_copy_to_host_async(void *dst_host, const void *src_device, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_host || !src_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_to_host_async: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return MVFX_OK;
}

int mvfx_copy_device_to_device_async(void *dst_device, const void *src_device, size_t bytes, mvfx_stream stream)
{
    if (bytes == 0) return MVFX_OK;
    if (!dst_device || !src_device)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "copy_device_to_device_async: NULL pointer");
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    MVFX_HIP_TRY(hipMemcpyAsync(dst_device, src_device, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return MVFX_OK;
}

int mvfx_thread_set_options(uint32_t options)
{
    const uint32_t known = MVFX_OPT_NONTEMPORAL | MVFX_OPT_HSV_LITERAL | MVFX_OPT_HSV_FORCE_FAST | MVFX_OPT_HSV_VALU_UNORM |
                           MVFX_OPT_LUT_PLACEMENT_MASK | MVFX_OPT_SSIM_F64 | MVFX_OPT_LUT_WG_WINDOW | MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY | MVFX_OPT_DIRECT_UNORDERED;
    if (options & ~known)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "thread options 0x%x: unknown bits 0x%x", options, options & ~known);
    if ((options & MVFX_OPT_HSV_LITERAL) && (options & MVFX_OPT_HSV_FORCE_FAST))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "thread options: MVFX_OPT_HSV_LITERAL and MVFX_OPT_HSV_FORCE_FAST exclude each other");
    // (every value of the three-bit field is a placement since round 5: 7 = round 4's per-wave windows)
    t_options = options;
    return MVFX_OK;
}

uint32_t mvfx_thread_options(void) { return t_options; }

} // extern "C"
