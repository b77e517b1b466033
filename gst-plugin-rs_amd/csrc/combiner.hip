// Launch combiner: the elements keep their contract -- one call per buffer (hsvfilter/imp.rs:322-326) -- yet the frames that the
// streaming threads of one process hand in at about the same time leave as ONE batched launch (blockIdx.z = frame, the settings
// of every frame in the kernel arguments).  A 4K frame is ~11.5 us of GPU work inside a 16-frame launch and ~16 us as a launch of
// its own; 16 threads x own stream x single-frame launches reach 0.66 of the HBM peak where the batched launch reaches 0.72
// (profiles/r3/bench_driver_command.json: every kernel boundary costs ~1-2 us of whole-chip time).
//
//   caller (a streaming thread, its own HIP stream S):
//       record event R on S                      -- everything the caller enqueued before (the fence wait of the buffer) orders first
//       enqueue {frame, settings, R}, wake the submitter, sleep until the frame's batch has been launched
//       make S wait for the batch's event D      -- whatever the caller enqueues next on S (its fence record, the next element's
//                                                   kernel) orders behind the batch: same ordering as the single-frame call
//   submitter (one thread per device, its own stream C):
//       wait for the first request; then until 16 requests are there, one of every stream that submitted in the last 2 ms, or the
//       window (MVFX_COMBINE_WINDOW_US, default 40) is over;  C waits for every R;  one launch for the frames that share geometry and format
//       (others: the next round);  record D on C;  wake the callers.
//
// The host never waits for the GPU in here.  Latency added per buffer: at most the window, in a busy process the time the other
// streams take to hand in their frames (microseconds).
#include "mvfx_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace mvfx {
namespace {

struct DoneSlot {
    hipEvent_t event = nullptr;
    int waiters = 0; // callers that still have to make their stream wait for `event` (guarded by Combiner::m)
};

struct Request {
    mvfx_frame frame;
    mvfx_hsvfilter_settings settings;
    hipEvent_t ready;
    uint32_t options;       // the caller's thread options (cache policy)
    bool launched = false;
    int status = MVFX_OK;
    char error[256] = "";
    DoneSlot *done = nullptr;
};

class Combiner {
public:
    explicit Combiner(int device) : device_(device)
    {
        if (const char *e = getenv("MVFX_COMBINE_WINDOW_US")) window_us_ = std::max(atoi(e), 0);
        worker_ = std::thread([this] { run(); });
    }
    ~Combiner()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_work_.notify_all();
        if (worker_.joinable()) worker_.join();
    }

    int submit(const mvfx_frame *frame, const mvfx_hsvfilter_settings *settings, hipStream_t caller_stream)
    {
        // the caller's ordering point: one event per calling thread and device, re-recorded per call (a record replaces the
        // previous one only after its batch has been launched, i.e. after the submitter's stream took its wait)
        struct ReadyEvents { // destroyed with the calling thread (GStreamer streaming threads come and go)
            std::map<int, hipEvent_t> by_device;
            ~ReadyEvents() { for (auto &kv : by_device) if (kv.second) (void)hipEventDestroy(kv.second); }
        };
        thread_local ReadyEvents t_ready;
        hipEvent_t &ready = t_ready.by_device[device_];
        if (!ready)
            MVFX_HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        MVFX_HIP_TRY(hipEventRecord(ready, caller_stream));
        Request req;
        req.frame = *frame;
        req.settings = *settings;
        req.ready = ready;
        req.options = thread_options();
        DoneSlot *done = nullptr;
        const auto t_in = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(m_);
            queue_.push_back(&req);
            seen_[std::this_thread::get_id()] = std::chrono::steady_clock::now();
            cv_work_.notify_one();
            cv_done_.wait(lk, [&] { return req.launched; });
            done = req.done;
        }
        wait_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_in).count();
        int rc = req.status;
        if (rc != MVFX_OK) fail(rc, "%s", req.error);
        if (done) {
            const hipError_t e = hipStreamWaitEvent(caller_stream, done->event, 0);
            {
                std::lock_guard<std::mutex> g(m_);
                done->waiters--;
            }
            if (e != hipSuccess && rc == MVFX_OK) rc = fail(MVFX_ERR_DEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
        }
        return rc;
    }

    void stats(uint64_t *batches, uint64_t *frames, double *avg_wait_us) const
    {
        *batches = batches_.load();
        *frames = frames_.load();
        if (avg_wait_us) *avg_wait_us = *frames ? (double)wait_ns_.load() / 1e3 / (double)*frames : 0.0;
    }

private:
    void run()
    {
        (void)hipSetDevice(device_);
        hipStream_t stream = nullptr;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) stream = nullptr;
        std::vector<DoneSlot> ring(64);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (stop_ && queue_.empty()) break;
            // collect: until the batch is full, every stream that has been handing in frames lately (the last 2 ms) has one in the
            // queue, or the window is over.  A lone stream -- or a 30 fps live one -- never waits: it is the only recent caller.
            const auto now = std::chrono::steady_clock::now();
            size_t expected = 0;
            for (auto it = seen_.begin(); it != seen_.end();) {
                if (now - it->second > std::chrono::milliseconds(2)) it = seen_.erase(it);
                else { ++expected; ++it; }
            }
            expected = std::min<size_t>(std::max<size_t>(expected, 1), kMaxCombine);
            const auto deadline = now + std::chrono::microseconds(window_us_);
            cv_work_.wait_until(lk, deadline, [&] { return stop_ || queue_.size() >= expected; });
            // frames that can share the first one's launch
            std::vector<Request *> batch;
            const Request *first = queue_.front();
            for (auto it = queue_.begin(); it != queue_.end() && batch.size() < (size_t)kMaxCombine;) {
                Request *r = *it;
                if (r->frame.width == first->frame.width && r->frame.height == first->frame.height && r->frame.stride == first->frame.stride &&
                    r->frame.format == first->frame.format && r->options == first->options) {
                    batch.push_back(r);
                    it = queue_.erase(it);
                } else {
                    ++it;
                }
            }
            DoneSlot *slot = nullptr;
            for (DoneSlot &s : ring)
                if (s.waiters == 0) { slot = &s; break; }
            // (64 slots, at most kMaxCombine callers hold one each: a free one always exists)
            slot->waiters = (int)batch.size();
            lk.unlock();

            int rc = MVFX_OK;
            char error[256] = "";
            if (!slot->event && hipEventCreateWithFlags(&slot->event, hipEventDisableTiming) != hipSuccess) rc = MVFX_ERR_DEVICE;
            std::vector<mvfx_frame> frames(batch.size());
            std::vector<mvfx_hsvfilter_settings> settings(batch.size());
            for (size_t i = 0; i < batch.size() && rc == MVFX_OK; i++) {
                frames[i] = batch[i]->frame;
                settings[i] = batch[i]->settings;
                if (hipStreamWaitEvent(stream, batch[i]->ready, 0) != hipSuccess) rc = MVFX_ERR_DEVICE;
            }
            if (rc == MVFX_OK) {
                (void)mvfx_thread_set_options(batch[0]->options);
                rc = mvfx_hsvfilter_transform_frames_ip_settings(frames.data(), (uint32_t)frames.size(), settings.data(), stream);
                if (rc != MVFX_OK) snprintf(error, sizeof(error), "%s", mvfx_last_error());
            } else {
                snprintf(error, sizeof(error), "launch combiner: a HIP event call failed: %s", hipGetErrorString(hipGetLastError()));
            }
            if (slot->event && hipEventRecord(slot->event, stream) != hipSuccess && rc == MVFX_OK) rc = MVFX_ERR_DEVICE;
            batches_++;
            frames_ += batch.size();

            lk.lock();
            for (Request *r : batch) {
                r->status = rc;
                if (rc != MVFX_OK) snprintf(r->error, sizeof(r->error), "%s", error);
                r->done = slot;
                r->launched = true;
            }
            cv_done_.notify_all();
        }
        lk.unlock();
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        for (DoneSlot &s : ring)
            if (s.event) (void)hipEventDestroy(s.event);
    }

    static constexpr int kMaxCombine = 16;
    const int device_;
    int window_us_ = 40;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Request *> queue_;
    std::map<std::thread::id, std::chrono::steady_clock::time_point> seen_; // callers and when they last submitted
    bool stop_ = false;
    std::thread worker_;
    std::atomic<uint64_t> batches_{0}, frames_{0}, wait_ns_{0}; // wait: submit -> launch enqueued, summed over the frames
};

std::mutex g_combiners_lock;
std::map<int, Combiner *> g_combiners; // one per device, for the life of the process (the worker thread parks on its condition variable)

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
                    fprintf(stderr, "mvfx combiner device %d: %llu launches for %llu frames (%.2f frames per launch), %.1f us from submit to launch per frame\n",
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
