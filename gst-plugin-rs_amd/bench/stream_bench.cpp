// libmvfxbench.so -- measurement harness for the launch model the elements really use: every
// stream has its own host thread (GStreamer: one streaming thread per stream) with its own HIP
// stream (mvfx_thread_stream) and calls the SINGLE-frame C-ABI entry point once per buffer
// (hsvfilter/imp.rs:322-326: one transform_frame_ip per buffer).  Python threads would put the
// GIL between the launches, so the threads live here; bench.py only starts the run and reads the
// clock values back.  Nothing in here computes pixels and the product library does not link it.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "mi355vfx.h"

namespace {

struct SpinBarrier {
    std::atomic<uint32_t> arrived{0};
    std::atomic<uint32_t> generation{0};
    uint32_t n;
    explicit SpinBarrier(uint32_t n_) : n(n_) {}
    void wait()
    {
        const uint32_t gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
        } else {
            while (generation.load(std::memory_order_acquire) == gen) std::this_thread::yield();
        }
    }
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

// n_threads host threads; thread t filters frames[t*frames_per_thread + (i % frames_per_thread)] for
// i in [0, warmup + launches), one mvfx_hsvfilter_transform_frame_ip per frame on its own stream, no
// synchronisation between launches (the device-memory element hands the buffer on with an event).
// Timed region (repeated `reps` times back to back, each bracketed by barriers): all threads released together after the
// previous work has drained; ends when the last thread's stream has drained.  seconds_out[r] = wall time of repetition r;
// thread_seconds[t] = thread t's own span in the last repetition.
//
// `batch` frames per launch (mvfx_hsvfilter_transform_frames_ip; 1 = the single-frame entry point): thread t walks its
// frames in groups of `batch` (frames_per_thread must be a multiple of it) -- the "few threads, each batching the streams it
// owns" model between the two extremes bench.py reports.
//
// `warm_frames` (may be NULL): warm_frames_per_thread scratch frames per thread for the warm-up launches, so that the timed
// launches start on frames no kernel has touched (frames that went through the filter many times are low-entropy, the chip
// draws less power on them and clocks higher: profiles/r2/exp_content_power.txt).  NULL = warm up on the timed frames.
//
// batch == 0: the launch combiner -- every thread calls the single-frame mvfx_hsvfilter_transform_frame_ip_combined and the
// library's submitter thread coalesces the frames of all threads into batched launches.
//
// streams_per_thread (mvfxbench_hsvfilter_streams_rot): every thread rotates its launches over that many private HIP streams
// (mvfx_thread_stream_n): consecutive frames of one video stream are independent, so the kernels of one thread may overlap.
static uint32_t g_streams_per_thread = 1;
// MVFXBENCH_SPLIT=n (round 4 experiment): every single-frame call becomes n calls on n horizontal bands of the frame, band p of call i
// on stream (i * n + p) % streams_per_thread -- what an element could do per buffer to shorten the fill and drain of a one-frame launch
static uint32_t split_bands()
{
    const char *e = getenv("MVFXBENCH_SPLIT");
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1u : (n > 8 ? 8u : (uint32_t)n);
}

int mvfxbench_hsvfilter_streams_warm(int device, uint32_t n_threads, uint32_t warmup, uint32_t launches, uint32_t reps,
                                     const mvfx_frame *frames, uint32_t frames_per_thread, uint32_t batch,
                                     const mvfx_frame *warm_frames, uint32_t warm_frames_per_thread,
                                     const mvfx_hsvfilter_settings *settings, uint32_t options, double *seconds_out,
                                     double *thread_seconds)
{
    const uint32_t rot = g_streams_per_thread ? g_streams_per_thread : 1;
    // batch == 0xFFFFFFFF: the combiner's fenced entry (no caller stream: the frame's previous launch's event in, a new event out)
    const bool fenced = batch == 0xFFFFFFFFu;
    const bool combined = batch == 0;
    if (combined || fenced) batch = 1;
    if (!frames || !settings || !seconds_out || n_threads == 0 || frames_per_thread == 0 || reps == 0 ||
        frames_per_thread % batch != 0 || (warm_frames && (warm_frames_per_thread == 0 || warm_frames_per_thread % batch != 0)))
        return MVFX_ERR_INVALID_ARGUMENT;
    const uint32_t groups = frames_per_thread / batch;
    const uint32_t split = split_bands();
    auto launch_in = [&](const mvfx_frame *mine, uint32_t n_groups, uint32_t i, mvfx_stream st) {
        const mvfx_frame *f = mine + (size_t)(i % n_groups) * batch;
        if (combined) return mvfx_hsvfilter_transform_frame_ip_combined(f, settings, st);
        if (split > 1 && batch == 1) {
            int rc = MVFX_OK;
            for (uint32_t p = 0; p < split && rc == MVFX_OK; p++) {
                mvfx_frame band = *f;
                const uint32_t r0 = (uint32_t)((uint64_t)f->height * p / split), r1 = (uint32_t)((uint64_t)f->height * (p + 1) / split);
                band.data = static_cast<uint8_t *>(f->data) + (size_t)r0 * f->stride;
                band.height = r1 - r0;
                rc = mvfx_hsvfilter_transform_frame_ip(&band, settings, mvfx_thread_stream_n((i * split + p) % rot));
            }
            return rc;
        }
        return batch == 1 ? mvfx_hsvfilter_transform_frame_ip(f, settings, st) : mvfx_hsvfilter_transform_frames_ip(f, batch, settings, st);
    };
    auto launch = [&](const mvfx_frame *mine, uint32_t i, mvfx_stream st) { return launch_in(mine, groups, i, st); };
    SpinBarrier ready(n_threads + 1), go(n_threads + 1), done(n_threads + 1);
    std::vector<int> status(n_threads, MVFX_OK);
    std::vector<double> span(n_threads, 0.0);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; t++) {
        pool.emplace_back([&, t] {
            int rc = mvfx_set_device(device);
            if (rc == MVFX_OK) rc = mvfx_thread_set_options(options);
            mvfx_stream sts[4];
            for (uint32_t k = 0; k < 4; k++) sts[k] = mvfx_thread_stream_n(k % rot);
            const mvfx_frame *mine = frames + (size_t)t * frames_per_thread;
            const mvfx_frame *warm = warm_frames ? warm_frames + (size_t)t * warm_frames_per_thread : mine;
            const uint32_t warm_groups = warm_frames ? warm_frames_per_thread / batch : groups;
            auto sync_all = [&]() { int e = MVFX_OK; for (uint32_t k = 0; k < rot && k < 4 && e == MVFX_OK; k++) e = mvfx_stream_synchronize(sts[k]); return e; };
            std::vector<mvfx_event> fence(frames_per_thread + (warm_frames ? warm_frames_per_thread : 0), nullptr); // per frame: its last launch
            auto launch_fenced = [&](const mvfx_frame *set, uint32_t n_frames, uint32_t base, uint32_t i) {
                const uint32_t k = i % n_frames;
                return mvfx_hsvfilter_transform_frame_ip_fenced(set + k, settings, fence[base + k], &fence[base + k]);
            };
            auto sync_fences = [&]() { int e = MVFX_OK; for (mvfx_event ev : fence) if (ev && e == MVFX_OK) e = mvfx_event_synchronize(ev); return e; };
            for (uint32_t i = 0; i < warmup && rc == MVFX_OK; i++)
                rc = fenced ? launch_fenced(warm, warm_frames ? warm_frames_per_thread : frames_per_thread, warm_frames ? frames_per_thread : 0, i)
                            : launch_in(warm, warm_groups, i, sts[i % rot % 4]);
            if (rc == MVFX_OK) rc = fenced ? sync_fences() : sync_all();
            ready.wait();
            for (uint32_t r = 0; r < reps; r++) {
                go.wait();
                const double t0 = now_s();
                for (uint32_t i = 0; i < launches && rc == MVFX_OK; i++)
                    rc = fenced ? launch_fenced(mine, frames_per_thread, 0, r * launches + i) : launch(mine, r * launches + i, sts[i % rot % 4]);
                if (rc == MVFX_OK) rc = fenced ? sync_fences() : sync_all();
                span[t] = now_s() - t0;
                done.wait();
            }
            status[t] = rc;
        });
    }
    ready.wait();
    for (uint32_t r = 0; r < reps; r++) {
        const double t0 = now_s();
        go.wait();
        done.wait();
        seconds_out[r] = now_s() - t0;
    }
    for (std::thread &th : pool) th.join();
    for (uint32_t t = 0; t < n_threads; t++) {
        if (thread_seconds) thread_seconds[t] = span[t];
        if (status[t] != MVFX_OK) return status[t];
    }
    return MVFX_OK;
}

int mvfxbench_hsvfilter_streams_rot(int device, uint32_t n_threads, uint32_t streams_per_thread, uint32_t warmup, uint32_t launches, uint32_t reps,
                                    const mvfx_frame *frames, uint32_t frames_per_thread, const mvfx_frame *warm_frames,
                                    uint32_t warm_frames_per_thread, const mvfx_hsvfilter_settings *settings, uint32_t options,
                                    double *seconds_out, double *thread_seconds)
{
    g_streams_per_thread = streams_per_thread < 1 ? 1 : (streams_per_thread > 4 ? 4 : streams_per_thread);
    const int rc = mvfxbench_hsvfilter_streams_warm(device, n_threads, warmup, launches, reps, frames, frames_per_thread, 1, warm_frames,
                                                    warm_frames_per_thread, settings, options, seconds_out, thread_seconds);
    g_streams_per_thread = 1;
    return rc;
}

// `batch` frames per launch AND the launches rotated over `streams_per_thread` private streams (round 4: what an element could reach by
// holding a few buffers back and launching them together)
int mvfxbench_hsvfilter_streams_rot_batched(int device, uint32_t n_threads, uint32_t streams_per_thread, uint32_t warmup, uint32_t launches,
                                            uint32_t reps, const mvfx_frame *frames, uint32_t frames_per_thread, uint32_t batch,
                                            const mvfx_hsvfilter_settings *settings, uint32_t options, double *seconds_out, double *thread_seconds)
{
    g_streams_per_thread = streams_per_thread < 1 ? 1 : (streams_per_thread > 4 ? 4 : streams_per_thread);
    const int rc = mvfxbench_hsvfilter_streams_warm(device, n_threads, warmup, launches, reps, frames, frames_per_thread, batch, nullptr, 0,
                                                    settings, options, seconds_out, thread_seconds);
    g_streams_per_thread = 1;
    return rc;
}

int mvfxbench_hsvfilter_streams_batched(int device, uint32_t n_threads, uint32_t warmup, uint32_t launches, uint32_t reps,
                                        const mvfx_frame *frames, uint32_t frames_per_thread, uint32_t batch,
                                        const mvfx_hsvfilter_settings *settings, uint32_t options, double *seconds_out,
                                        double *thread_seconds)
{
    return mvfxbench_hsvfilter_streams_warm(device, n_threads, warmup, launches, reps, frames, frames_per_thread, batch, nullptr, 0,
                                            settings, options, seconds_out, thread_seconds);
}

int mvfxbench_hsvfilter_streams(int device, uint32_t n_threads, uint32_t warmup, uint32_t launches, uint32_t reps, const mvfx_frame *frames,
                                uint32_t frames_per_thread, const mvfx_hsvfilter_settings *settings, uint32_t options,
                                double *seconds_out, double *thread_seconds)
{
    return mvfxbench_hsvfilter_streams_batched(device, n_threads, warmup, launches, reps, frames, frames_per_thread, 1, settings, options,
                                               seconds_out, thread_seconds);
}

// The same launch model for colorlut: thread t grades in_frames[t*frames_per_thread + i] into out_frames[...] with the shared LUT
// handle, one mvfx_colorlut_transform_frame per frame on its own stream.
int mvfxbench_colorlut_streams(int device, uint32_t n_threads, uint32_t warmup, uint32_t launches, uint32_t reps, mvfx_cube_lut *lut,
                               const mvfx_frame *in_frames, const mvfx_frame *out_frames, uint32_t frames_per_thread, uint32_t options,
                               double *seconds_out, double *thread_seconds)
{
    if (!lut || !in_frames || !out_frames || !seconds_out || n_threads == 0 || frames_per_thread == 0 || reps == 0)
        return MVFX_ERR_INVALID_ARGUMENT;
    SpinBarrier ready(n_threads + 1), go(n_threads + 1), done(n_threads + 1);
    std::vector<int> status(n_threads, MVFX_OK);
    std::vector<double> span(n_threads, 0.0);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; t++) {
        pool.emplace_back([&, t] {
            int rc = mvfx_set_device(device);
            if (rc == MVFX_OK) rc = mvfx_thread_set_options(options);
            mvfx_stream st = mvfx_thread_stream();
            const mvfx_frame *in = in_frames + (size_t)t * frames_per_thread, *out = out_frames + (size_t)t * frames_per_thread;
            for (uint32_t i = 0; i < warmup && rc == MVFX_OK; i++)
                rc = mvfx_colorlut_transform_frame(lut, &in[i % frames_per_thread], &out[i % frames_per_thread], st);
            if (rc == MVFX_OK) rc = mvfx_stream_synchronize(st);
            ready.wait();
            for (uint32_t r = 0; r < reps; r++) {
                go.wait();
                const double t0 = now_s();
                for (uint32_t i = 0; i < launches && rc == MVFX_OK; i++)
                    rc = mvfx_colorlut_transform_frame(lut, &in[(warmup + i) % frames_per_thread], &out[(warmup + i) % frames_per_thread], st);
                if (rc == MVFX_OK) rc = mvfx_stream_synchronize(st);
                span[t] = now_s() - t0;
                done.wait();
            }
            status[t] = rc;
        });
    }
    ready.wait();
    for (uint32_t r = 0; r < reps; r++) {
        const double t0 = now_s();
        go.wait();
        done.wait();
        seconds_out[r] = now_s() - t0;
    }
    for (std::thread &th : pool) th.join();
    for (uint32_t t = 0; t < n_threads; t++) {
        if (thread_seconds) thread_seconds[t] = span[t];
        if (status[t] != MVFX_OK) return status[t];
    }
    return MVFX_OK;
}

// The element's contract through the direct-dispatch lane (round 6, MVFX_OPT_DIRECT_DISPATCH): ONE thread, one single-frame call per buffer, every
// frame with its own fence -- an event out of a ring of 32, re-used when its frame has finished, as the element layer's fence pool does -- and
// nothing enqueued on a stream.  *direct_out = how many of the timed launches the lane took (0: no lane on this box, the figure is the stream's).
int mvfxbench_hsvfilter_direct(int device, uint32_t warmup, uint32_t launches, uint32_t reps, const mvfx_frame *frames, uint32_t n_frames,
                               const mvfx_hsvfilter_settings *settings, uint32_t options, double *seconds_out, uint64_t *direct_out)
{
    if (!frames || !settings || !seconds_out || n_frames == 0 || reps == 0) return MVFX_ERR_INVALID_ARGUMENT;
    int rc = MVFX_OK;
    uint64_t direct = 0;
    std::thread th([&] {
        rc = mvfx_set_device(device);
        constexpr uint32_t kEvents = 32;
        mvfx_event ev[kEvents] = {};
        for (uint32_t k = 0; k < kEvents && rc == MVFX_OK; k++) rc = mvfx_event_create(&ev[k]);
        // consecutive frames alternate between the thread's first two streams, as the elements do (mvfx_element_stream: by frame number): the
        // stream decides which of the lane's two in-order queues a dispatch takes
        mvfx_stream sts[2] = {mvfx_thread_stream_n(0), mvfx_thread_stream_n(1)};
        uint64_t n = 0;
        auto one = [&](uint32_t i, bool count) {
            mvfx_stream st = sts[n & 1];
            mvfx_event e = ev[n % kEvents];
            if (n >= kEvents) rc = mvfx_event_synchronize(e);
            n++;
            if (rc != MVFX_OK) return;
            mvfx_thread_set_options(options | MVFX_OPT_DIRECT_DISPATCH);
            mvfx_thread_set_completion_event(e);
            rc = mvfx_hsvfilter_transform_frame_ip(&frames[i % n_frames], settings, st);
            if (mvfx_thread_clear_completion_event() <= 0 && rc == MVFX_OK) rc = mvfx_event_record(e, st);
            if (count && mvfx_event_is_direct(e)) direct++;
        };
        auto drain = [&] { for (uint32_t k = 0; k < kEvents && rc == MVFX_OK; k++) if (n > k) rc = mvfx_event_synchronize(ev[k]); };
        for (uint32_t i = 0; i < warmup && rc == MVFX_OK; i++) one(i, false);
        drain();
        for (uint32_t r = 0; r < reps && rc == MVFX_OK; r++) {
            const double t0 = now_s();
            for (uint32_t i = 0; i < launches && rc == MVFX_OK; i++) one(r * launches + i, true);
            drain();
            seconds_out[r] = now_s() - t0;
        }
        mvfx_thread_set_options(0);
        for (uint32_t k = 0; k < kEvents; k++) if (ev[k]) mvfx_event_destroy(ev[k]);
    });
    th.join();
    if (direct_out) *direct_out = direct;
    return rc;
}

// The same for colorlut: frame i goes from ins[i % n_frames] to outs[i % n_frames], one call per frame, one fence per frame; without
// MVFX_OPT_DIRECT_DISPATCH in `options` the calls are the ordinary single-frame launches on the thread's two streams (the element's round-5 path).
int mvfxbench_colorlut_direct(int device, uint32_t warmup, uint32_t launches, uint32_t reps, mvfx_cube_lut *lut, const mvfx_frame *ins, const mvfx_frame *outs,
                              uint32_t n_frames, uint32_t options, double *seconds_out, uint64_t *direct_out)
{
    if (!lut || !ins || !outs || !seconds_out || n_frames == 0 || reps == 0) return MVFX_ERR_INVALID_ARGUMENT;
    int rc = MVFX_OK;
    uint64_t direct = 0;
    std::thread th([&] {
        rc = mvfx_set_device(device);
        constexpr uint32_t kEvents = 32;
        mvfx_event ev[kEvents] = {};
        for (uint32_t k = 0; k < kEvents && rc == MVFX_OK; k++) rc = mvfx_event_create(&ev[k]);
        mvfx_stream sts[2] = {mvfx_thread_stream_n(0), mvfx_thread_stream_n(1)};
        uint64_t n = 0;
        auto one = [&](uint32_t i, bool count) {
            mvfx_stream st = sts[n & 1];
            mvfx_event e = ev[n % kEvents];
            if (n >= kEvents) rc = mvfx_event_synchronize(e);
            n++;
            if (rc != MVFX_OK) return;
            mvfx_thread_set_options(options);
            mvfx_thread_set_completion_event(e);
            rc = mvfx_colorlut_transform_frame(lut, &ins[i % n_frames], &outs[i % n_frames], st);
            if (mvfx_thread_clear_completion_event() <= 0 && rc == MVFX_OK) rc = mvfx_event_record(e, st);
            if (count && mvfx_event_is_direct(e)) direct++;
        };
        auto drain = [&] { for (uint32_t k = 0; k < kEvents && rc == MVFX_OK; k++) if (n > k) rc = mvfx_event_synchronize(ev[k]); };
        for (uint32_t i = 0; i < warmup && rc == MVFX_OK; i++) one(i, false);
        drain();
        for (uint32_t r = 0; r < reps && rc == MVFX_OK; r++) {
            const double t0 = now_s();
            for (uint32_t i = 0; i < launches && rc == MVFX_OK; i++) one(r * launches + i, true);
            drain();
            seconds_out[r] = now_s() - t0;
        }
        mvfx_thread_set_options(0);
        for (uint32_t k = 0; k < kEvents; k++) if (ev[k]) mvfx_event_destroy(ev[k]);
    });
    th.join();
    if (direct_out) *direct_out = direct;
    return rc;
}

// Several threads share the lane, each with a SMALL ring of fences (as the element layer's fence pool under a small buffer pool): `threads` threads,
// each making `launches` one-frame hsvfilter calls through the lane on its own frames, re-using `events_per_thread` events (waiting for an event's
// previous frame before re-using it).  What it is for: the lane's argument-slot ring (1024 slots) wraps many times and every slot's "last user" is
// one of a handful of signals that are armed again and again -- the constellation in which round 6's first lane deadlocked two streaming threads.
// Returns MVFX_OK when every thread has finished; a deadlock never returns (the test has a timeout).  *direct_out: launches the lane took.
int mvfxbench_lane_threads(int device, uint32_t threads, uint32_t events_per_thread, uint32_t launches, const mvfx_frame *frames, uint32_t frames_per_thread,
                           const mvfx_hsvfilter_settings *settings, uint64_t *direct_out)
{
    if (!frames || !settings || threads == 0 || events_per_thread == 0 || events_per_thread > 64 || frames_per_thread == 0) return MVFX_ERR_INVALID_ARGUMENT;
    std::vector<int> rcs(threads, MVFX_OK);
    std::vector<uint64_t> took(threads, 0);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            int rc = mvfx_set_device(device);
            mvfx_event ev[64] = {};
            for (uint32_t k = 0; k < events_per_thread && rc == MVFX_OK; k++) rc = mvfx_event_create(&ev[k]);
            mvfx_stream sts[2] = {mvfx_thread_stream_n(0), mvfx_thread_stream_n(1)};
            for (uint32_t i = 0; i < launches && rc == MVFX_OK; i++) {
                mvfx_event e = ev[i % events_per_thread];
                if (i >= events_per_thread) rc = mvfx_event_synchronize(e);
                if (rc != MVFX_OK) break;
                mvfx_thread_set_options(MVFX_OPT_DIRECT_DISPATCH);
                mvfx_thread_set_completion_event(e);
                rc = mvfx_hsvfilter_transform_frame_ip(&frames[t * frames_per_thread + i % frames_per_thread], settings, sts[i & 1]);
                if (mvfx_thread_clear_completion_event() <= 0 && rc == MVFX_OK) rc = mvfx_event_record(e, sts[i & 1]);
                if (mvfx_event_is_direct(e)) took[t]++;
            }
            for (uint32_t k = 0; k < events_per_thread; k++) if (ev[k]) { if (rc == MVFX_OK) rc = mvfx_event_synchronize(ev[k]); mvfx_event_destroy(ev[k]); }
            mvfx_thread_set_options(0);
            rcs[t] = rc;
        });
    for (auto &th : pool) th.join();
    uint64_t all = 0;
    for (uint32_t t = 0; t < threads; t++) all += took[t];
    if (direct_out) *direct_out = all;
    for (int rc : rcs) if (rc != MVFX_OK) return rc;
    return MVFX_OK;
}

} // extern "C"
