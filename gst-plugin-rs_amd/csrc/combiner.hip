// Launch combiner: the elements keep their contract -- one call per buffer (hsvfilter/imp.rs:322-326) -- yet the frames that the
// streaming threads of one process hand in at about the same time leave as ONE batched launch (blockIdx.z = frame, the settings
// of every frame in the kernel arguments).  A 4K frame is ~11.5 us of GPU work inside a 16-frame launch and ~16 us as a launch of
// its own; 16 threads x own stream x single-frame launches reach 0.66 of the HBM peak where the batched launch reaches 0.72: every
// kernel boundary costs ~1-2 us of whole-chip time.
//
// No extra thread.  The first caller that finds no open batch becomes its LEADER; callers that arrive while it collects join as
// followers:
//   follower (its own stream S): record event R on S unless S is idle (everything the caller enqueued before orders first), put
//       {frame, settings, R} into the open batch, spin until the leader has launched, then make S wait for the batch's event D
//       (whatever the caller enqueues next on S -- its fence record, the next element's kernel -- orders behind the batch);
//   leader (its own stream L): spins until 16 frames are in, every stream that submitted in the last 2 ms has one in, or the
//       window (MVFX_COMBINE_WINDOW_US, default 30) is over; closes the batch; L waits for every follower's R; ONE launch on L
//       for all frames (frames that do not share the first one's geometry / format / cache policy are not admitted: they open the
//       next batch); records D on L; releases the followers.
// The host never waits for the GPU in here, and a lone stream is never held back (it is the only recent caller: its batch closes at
// once).  Round 3 history: the first version handed the frames to a submitter thread through a condition variable -- two thread
// hand-offs and a mutex convoy of 16 woken callers per batch: 46.6 k fps where plain per-stream launches do 79.6 k
// (profiles/r3/combiner_first_version.txt); leader / follower with short spins removed the hand-offs.
#include "mvfx_internal.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace mvfx {
namespace {

constexpr int kMaxCombine = 16;
using Clock = std::chrono::steady_clock;

struct Request {
    mvfx_frame frame;
    mvfx_hsvfilter_settings settings;
    hipEvent_t ready; // stream mode: the caller's ordering point (nullptr: its stream was idle); fenced mode: the frame's fence (nullptr: none)
};

struct Batch {
    Request *req[kMaxCombine];
    std::atomic<int> n{0};
    std::atomic<int> launched{0}; // release-stored by the leader after the launch has been enqueued
    int waiters = 0;              // followers that still have to take `done` (guarded by Combiner::m)
    bool open = false;
    uint32_t options = 0;
    hipEvent_t done = nullptr;
    int status = MVFX_OK;
    char error[256] = "";
};

inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

class Combiner {
public:
    explicit Combiner(int device) : device_(device), ring_(256)
    {
        if (const char *e = getenv("MVFX_COMBINE_WINDOW_US")) window_us_ = std::max(atoi(e), 0);
    }

    // fenced == false: `stream` is the caller's stream (ordering through it, see the file comment).
    // fenced == true:  no caller stream at all.  `wait_for` is the frame's fence (an event its previous user recorded, or NULL), the
    //   launch runs on the combiner's OWN stream of this device -- every fenced launch of the device, in submission order -- and
    //   *done_out is the event behind it: the frame's new fence.  Nothing is enqueued on any other stream, so consecutive batches are
    //   separated by one kernel boundary only, like the batched entry point called from one thread.
    int submit(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings, hipStream_t stream, bool fenced = false,
               hipEvent_t wait_for = nullptr, hipEvent_t *done_out = nullptr)
    {
        const auto t_in = Clock::now();
        Request req{*frame, *settings, fenced ? wait_for : nullptr};
        const uint32_t options = thread_options() | (fenced ? 0x80000000u : 0u); // the two modes never share a batch
        if (fenced) {
            std::lock_guard<std::mutex> g(m_);
            if (!own_stream_ && hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking) != hipSuccess)
                return fail(MVFX_ERR_DEVICE, "launch combiner: no stream: %s", hipGetErrorString(hipGetLastError()));
            stream = own_stream_;
        }
        Batch *b = nullptr;
        bool leader = false;
        size_t expected = 1;
        {
            std::unique_lock<std::mutex> lk(m_);
            seen_[std::this_thread::get_id()] = t_in;
            Batch *o = open_;
            if (o && o->open && o->n.load(std::memory_order_relaxed) < kMaxCombine && o->options == options &&
                same_geometry(o->req[0]->frame, *frame)) {
                // follower: the ordering point of this stream, unless nothing is pending on it
                lk.unlock();
                if (!fenced && hipStreamQuery(stream) != hipSuccess) {
                    (void)hipGetLastError();
                    if (int rc = ready_event(&req.ready); rc != MVFX_OK) return rc;
                    MVFX_HIP_TRY(hipEventRecord(req.ready, stream));
                }
                lk.lock();
                o = open_; // the batch may have closed while the event was recorded
                if (o && o->open && o->n.load(std::memory_order_relaxed) < kMaxCombine && o->options == options &&
                    same_geometry(o->req[0]->frame, *frame)) {
                    b = o;
                    b->req[b->n.load(std::memory_order_relaxed)] = &req;
                    b->waiters++;
                    b->n.fetch_add(1, std::memory_order_release);
                }
            }
            if (!b) { // leader of a new batch (an open batch this frame cannot join stays open for its own leader)
                leader = true;
                for (Batch &s : ring_)
                    if (!s.open && s.waiters == 0 && (s.n.load(std::memory_order_relaxed) == 0 || s.launched.load(std::memory_order_relaxed))) { b = &s; break; }
                // (256 slots; a slot is busy only while one of at most 16 followers still has to take its event)
                if (!b) {
                    // every slot still has a follower that has not taken its event (256 slots: it has not been seen): the frame
                    // goes out on its own instead of failing the buffer -- a plain launch on the caller's stream; in fenced mode on
                    // the combiner's stream behind the frame's fence, with a fresh event as the new fence
                    lk.unlock();
                    if (!fenced) return mvfx_hsvfilter_transform_frame_ip(frame, settings, stream);
                    if (wait_for && hipStreamWaitEvent(stream, wait_for, 0) != hipSuccess)
                        return fail(MVFX_ERR_DEVICE, "launch combiner: hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
                    if (int rc1 = mvfx_hsvfilter_transform_frame_ip(frame, settings, stream); rc1 != MVFX_OK) return rc1;
                    if (done_out) {
                        hipEvent_t ev = nullptr;
                        if (int rc1 = ready_event(&ev); rc1 != MVFX_OK) return rc1;
                        MVFX_HIP_TRY(hipEventRecord(ev, stream));
                        *done_out = ev;
                    }
                    return MVFX_OK;
                }
                b->n.store(0, std::memory_order_relaxed);
                b->launched.store(0, std::memory_order_relaxed);
                b->options = options;
                b->status = MVFX_OK;
                b->req[0] = &req;
                b->n.store(1, std::memory_order_release);
                b->open = true;
                if (!open_ || !open_->open) open_ = b; // else: another geometry is collecting, this one goes alone
                const bool alone = open_ != b;
                for (auto it = seen_.begin(); it != seen_.end();) { // streams that handed in a frame in the last 2 ms
                    if (t_in - it->second > std::chrono::milliseconds(2)) it = seen_.erase(it);
                    else { ++expected; ++it; }
                }
                expected = alone ? 1 : std::min<size_t>(std::max<size_t>(expected - 1, 1), kMaxCombine);
            }
        }
        int rc = MVFX_OK;
        if (leader) {
            // collect: spin (the other streams are microseconds away), yield when the window is long
            const auto deadline = t_in + std::chrono::microseconds(window_us_);
            while ((size_t)b->n.load(std::memory_order_acquire) < expected && Clock::now() < deadline) cpu_relax();
            int n;
            {
                std::lock_guard<std::mutex> g(m_);
                b->open = false;
                if (open_ == b) open_ = nullptr;
                n = b->n.load(std::memory_order_acquire);
            }
            if (!b->done && hipEventCreateWithFlags(&b->done, hipEventDisableTiming) != hipSuccess) rc = MVFX_ERR_DEVICE;
            mvfx_frame frames[kMaxCombine];
            mvfx_hsvfilter_settings settings_all[kMaxCombine];
            for (int i = 0; i < n; i++) {
                frames[i] = b->req[i]->frame;
                settings_all[i] = b->req[i]->settings;
                if ((i > 0 || fenced) && b->req[i]->ready && rc == MVFX_OK && hipStreamWaitEvent(stream, b->req[i]->ready, 0) != hipSuccess)
                    rc = MVFX_ERR_DEVICE;
            }
            if (rc == MVFX_OK) {
                const uint32_t mine = thread_options();
                if (fenced) (void)mvfx_thread_set_options(b->options & 0x7fffffffu);
                rc = mvfx_hsvfilter_transform_frames_ip_settings(frames, (uint32_t)n, settings_all, stream);
                if (fenced) (void)mvfx_thread_set_options(mine);
            }
            else
                fail(rc, "launch combiner: a HIP event call failed: %s", hipGetErrorString(hipGetLastError()));
            if (rc != MVFX_OK) snprintf(b->error, sizeof(b->error), "%s", mvfx_last_error());
            if ((n > 1 || fenced) && b->done && hipEventRecord(b->done, stream) != hipSuccess && rc == MVFX_OK) rc = MVFX_ERR_DEVICE;
            if (fenced && done_out) *done_out = b->done;
            b->status = rc;
            batches_++;
            frames_ += (uint64_t)n;
            b->launched.store(1, std::memory_order_release);
        } else {
            // follower: the leader is busy launching; spin, then yield
            int spins = 0;
            while (!b->launched.load(std::memory_order_acquire)) {
                if (++spins < 4000) cpu_relax();
                else std::this_thread::yield();
            }
            rc = b->status;
            if (rc != MVFX_OK) fail(rc, "%s", b->error);
            if (fenced && done_out) *done_out = b->done;
            const hipError_t e = fenced ? hipSuccess : (b->done ? hipStreamWaitEvent(stream, b->done, 0) : hipErrorInvalidValue);
            {
                std::lock_guard<std::mutex> g(m_);
                b->waiters--;
            }
            if (e != hipSuccess && rc == MVFX_OK) rc = fail(MVFX_ERR_DEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
        }
        wait_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t_in).count();
        return rc;
    }

    void stats(uint64_t *batches, uint64_t *frames, double *avg_wait_us) const
    {
        *batches = batches_.load();
        *frames = frames_.load();
        if (avg_wait_us) *avg_wait_us = *frames ? (double)wait_ns_.load() / 1e3 / (double)*frames : 0.0;
    }

private:
    static bool same_geometry(const mvfx_frame &a, const mvfx_frame &b)
    {
        return a.width == b.width && a.height == b.height && a.stride == b.stride && a.format == b.format;
    }

    // one event per calling thread and device, re-recorded per call (the previous record has been waited for by then: the caller
    // returned from that submit only after its batch was launched)
    int ready_event(hipEvent_t *out)
    {
        // The events of exited threads are POOLED, never destroyed: GStreamer streaming threads come and go, sixteen of them leave at
        // end-of-stream at once, and concurrent teardown of HIP objects from exiting threads is what crashed the runtime with streams
        // (profiles/r3/stream_destroy_crash_backtrace.txt; capi_common.hip pools streams for the same reason).  The pool is leaked on
        // purpose (threads may exit after the static destructors ran).
        struct EventPool {
            std::mutex lock;
            std::map<int, std::vector<hipEvent_t>> idle;
        };
        static EventPool *pool = new EventPool;
        struct ReadyEvents {
            std::map<int, hipEvent_t> by_device;
            ~ReadyEvents()
            {
                std::lock_guard<std::mutex> g(pool->lock);
                for (auto &kv : by_device)
                    if (kv.second) pool->idle[kv.first].push_back(kv.second);
            }
        };
        thread_local ReadyEvents t_ready;
        hipEvent_t &ready = t_ready.by_device[device_];
        if (!ready) {
            std::lock_guard<std::mutex> g(pool->lock);
            std::vector<hipEvent_t> &idle = pool->idle[device_];
            if (!idle.empty()) {
                ready = idle.back();
                idle.pop_back();
            }
        }
        if (!ready)
            MVFX_HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        *out = ready;
        return MVFX_OK;
    }

    const int device_;
    int window_us_ = 30;
    std::mutex m_;
    std::vector<Batch> ring_;
    Batch *open_ = nullptr;
    hipStream_t own_stream_ = nullptr; // fenced mode: every launch of this device, in submission order
    std::map<std::thread::id, Clock::time_point> seen_; // callers and when they last submitted
    std::atomic<uint64_t> batches_{0}, frames_{0}, wait_ns_{0}; // wait: time inside submit, summed over the frames
};

std::mutex g_combiners_lock;
std::map<int, Combiner *> g_combiners; // one per device, for the life of the process

Combiner *combiner_for(int device)
{
    std::lock_guard<std::mutex> g(g_combiners_lock);
    Combiner *&c = g_combiners[device];
    if (!c) {
        c = new Combiner(device);
        static bool report = false;
        if (!report && getenv("MVFX_COMBINE_STATS")) { // gst-launch runs: the numbers at process exit
            report = true;
            atexit([] {
                for (auto &kv : g_combiners) {
                    uint64_t b = 0, f = 0;
                    double wait = 0.0;
                    kv.second->stats(&b, &f, &wait);
                    fprintf(stderr, "mvfx combiner device %d: %llu launches for %llu frames (%.2f frames per launch), %.1f us per call\n",
                            kv.first, (unsigned long long)b, (unsigned long long)f, b ? (double)f / (double)b : 0.0, wait);
                }
            });
        }
    }
    return c;
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_hsvfilter_transform_frame_ip_combined(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings, mvfx_stream stream)
{
    if (!frame || !settings)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: NULL frame or settings");
    if (int rc = check_packed_frame(frame, "hsvfilter"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    int device = 0;
    MVFX_HIP_TRY(hipGetDevice(&device));
    return combiner_for(device)->submit(frame, settings, as_stream(stream));
}

int mvfx_hsvfilter_transform_frame_ip_fenced(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings, mvfx_event wait_for,
                                             mvfx_event *done_out)
{
    if (!frame || !settings || !done_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "hsvfilter: NULL frame, settings or event output");
    *done_out = nullptr;
    if (int rc = check_packed_frame(frame, "hsvfilter"); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    int device = 0;
    MVFX_HIP_TRY(hipGetDevice(&device));
    hipEvent_t done = nullptr;
    const int rc = combiner_for(device)->submit(frame, settings, nullptr, true, reinterpret_cast<hipEvent_t>(wait_for), &done);
    *done_out = reinterpret_cast<mvfx_event>(done);
    return rc;
}

int mvfx_combiner_stats(int device, uint64_t *batches_out, uint64_t *frames_out)
{
    if (!batches_out || !frames_out) return fail(MVFX_ERR_INVALID_ARGUMENT, "combiner: NULL output");
    *batches_out = *frames_out = 0;
    std::lock_guard<std::mutex> g(g_combiners_lock);
    auto it = g_combiners.find(device);
    if (it != g_combiners.end()) it->second->stats(batches_out, frames_out, nullptr);
    return MVFX_OK;
}

double mvfx_combiner_average_wait_us(int device)
{
    uint64_t b = 0, f = 0;
    double wait = 0.0;
    std::lock_guard<std::mutex> g(g_combiners_lock);
    auto it = g_combiners.find(device);
    if (it != g_combiners.end()) it->second->stats(&b, &f, &wait);
    return wait;
}

} // extern "C"
