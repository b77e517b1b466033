// Pair launches on memory:HIPMemory buffers -- OPT-IN (MVFX_ELEMENT_PAIR=1 or 2; the default is one launch per buffer).
//
// One 4K frame per launch fills and drains the chip for 33-66 MB; two consecutive frames in one launch on two alternating streams
// close most of the distance to the batched entries (profiles/r4/element_path.txt).  With pairs on, the element contract stays one
// transform() call per buffer (hsvfilter/imp.rs:322, hsvdetector/imp.rs:422, colorlut/imp.rs:203): the call returns at once, the
// frame's kernel is HELD BACK, its blocks -- the input (out-of-place elements), which the upstream pool may hand out again, and the
// output, which the next element reads -- carry the element's deferred mark (mvfx_hip_memory_set_deferred), and the kernel leaves
// together with the next buffer's frame: or alone, the moment anybody looks at either block's fence (the next element's acquire, a
// CPU map, hipdownload, the source refilling a recycled block), when the geometry or the settings change, at EOS, on flush-start, in
// stop() -- and when no second buffer has come within one frame interval (the idle flush below).
//
// Why it is not the default (VERDICT r4 W6 / W-semantics): it only pays where nobody consumes the buffer (behind another device element
// every held-back frame is flushed alone by that element's first look), and the reference returns a frame's failure from that frame's
// own transform call.  What this file guarantees when it is switched on:
//   * a held-back frame whose launch fails is reported TWICE: posted on the bus (after every lock is released), and the element's next
//     transform() -- and every later one until stop() -- returns the failure as its flow return (GST_FLOW_ERROR), like the reference's
//     synchronous error would have ended the stream one buffer earlier;
//   * a frame never sits longer than one frame interval: a process-wide timer thread (mvfx_idle_arm) launches it alone;
//   * the marks are never left stale: after they are set the hold re-checks that the frame is still held.
//
// Holding back only pays when nobody looks: behind another device element every held-back frame is flushed alone by that element's
// first look, and the chain hsvfilter ! hsvdetector ! colorlut ran at 13.3 k instead of 17.3 k fps with all three holding
// (profiles/r4/element_pairs.txt).  So in mode 1 the hold watches itself: kMvfxPairStreak held-back frames in a row that somebody else's
// look flushed switch the element to a plain launch per buffer for the next kMvfxPairDirect buffers, then it tries again.  And a buffer
// whose INPUT block is still being written when it arrives (mvfx_hip_memory_busy: an upstream kernel or the source's copy in flight,
// or held back) is launched the plain way at once: the LAST element of that chain paired cost 22.8 k -> 14.8 k fps, because a pair
// waits for the second frame's whole upstream chain where two single frames pipeline on two streams.
#pragma once

#include "mvfx_gst_common.h"

#include <mutex>

struct MvfxPairHold {
    std::mutex lock;
    GstMemory *in_mem = NULL, *out_mem = NULL; // referenced while set; in_mem stays NULL for an in-place element (hsvfilter)
    mvfx_frame fi, fo;
    mvfx_stream st = NULL;                      // the stream a lone launch of the held-back frame goes on
    int device = -1;                            // ... and its device (the idle flush runs on a thread of its own)
    guint pair_no = 0;
    guint foreign_streak = 0, direct_left = 0;  // see above
    int failed_rc = MVFX_OK;                    // a held-back frame's launch failed: sticky until stop()
    gchar *failed_text = NULL;                  // ... and what the library said (the failure may have happened on another thread)
    ~MvfxPairHold() { g_free(failed_text); }
    guint64 idle_us = 10000;                    // one frame interval (mvfx_pair_set_interval)
    gint64 held_since = 0;                      // g_get_monotonic_time() of the hold
    guint64 n_buffers = 0, n_pairs = 0, n_singles = 0, n_direct = 0, n_idle = 0;
};
constexpr guint kMvfxPairStreak = 4, kMvfxPairDirect = 1024;
constexpr int MVFX_PAIR_NOT_TAKEN = 0x7fff0001; // mvfx_pair_submit: the caller launches this buffer itself, the plain way
constexpr int MVFX_PAIR_FAILED_EARLIER = 0x7fff0002; // mvfx_pair_submit: a held-back frame failed after its call had returned; the caller
                                                     // returns mvfx_pair_flow_error() -- this call carries that frame's error

// Memory references dropped AFTER the element's lock is released (declare it before the lock guard): the last reference frees the
// block, and freeing runs a stale mark's callback -- no callback of another element ever runs under this element's lock.
struct MvfxUnrefLater {
    GstMemory *m[4];
    int n = 0;
    void add(GstMemory *mem) { if (mem) m[n++] = mem; }
    ~MvfxUnrefLater() { for (int i = 0; i < n; i++) gst_memory_unref(m[i]); }
};

// An error message posted AFTER the element's lock is released (declare it before the lock guard): a bus sync handler may call back
// into the element (advisor r4: posting under the hold's lock can deadlock with a listener that touches this element's buffers).
struct MvfxPostLater {
    GstObject *element = NULL;
    gchar *text = NULL;
    int rc = MVFX_OK;
    void set(GstObject *e, int code) { if (!element) { element = e; rc = code; text = g_strdup(mvfx_last_error()); } }
    ~MvfxPostLater()
    {
        if (!element) return;
        GST_ELEMENT_ERROR(GST_ELEMENT(element), LIBRARY, FAILED, ("%s", text), ("held-back frame: mvfx status %d (%s); the element's next "
                          "transform returns the error", rc, mvfx_status_string(rc)));
        g_free(text);
    }
};

// Launches n (1 or 2) frames of the element on `st`; called with the hold's lock held, so it may read what the element stored next
// to the held-back frame (its settings).  In-place elements get in == out.
typedef int (*MvfxPairLaunch)(GstObject *element, const mvfx_frame *in, const mvfx_frame *out, uint32_t n, mvfx_stream st);
struct MvfxPairOps {
    MvfxPairLaunch launch;
    MvfxDeferredFlush looked_at; // registered on the held-back frame's blocks
    MvfxDeferredFlush idle;      // registered with the timer thread
};

// MVFX_ELEMENT_PAIR: 0 (default) = a launch per buffer, the reference's contract to the letter; 1 = hold back unless the element finds
// itself inside a chain (see above); 2 = always hold back (tests: the cross-thread flushes of neighbouring elements, all the time)
static inline int mvfx_pair_mode(void)
{
    static const int mode = g_getenv("MVFX_ELEMENT_PAIR") ? atoi(g_getenv("MVFX_ELEMENT_PAIR")) : 0;
    return mode;
}
static inline gboolean mvfx_pair_enabled(void) { return mvfx_pair_mode() != 0; }

// one frame interval of the negotiated caps (variable framerate: 10 ms), MVFX_PAIR_IDLE_US overrides (tests)
static inline void mvfx_pair_set_interval(MvfxPairHold *h, const GstVideoInfo *info)
{
    static const gint64 forced = g_getenv("MVFX_PAIR_IDLE_US") ? g_ascii_strtoll(g_getenv("MVFX_PAIR_IDLE_US"), NULL, 10) : 0;
    guint64 us = 10000;
    if (forced > 0) us = (guint64)forced;
    else if (info && GST_VIDEO_INFO_FPS_N(info) > 0 && GST_VIDEO_INFO_FPS_D(info) > 0)
        us = gst_util_uint64_scale_int(1000000, GST_VIDEO_INFO_FPS_D(info), GST_VIDEO_INFO_FPS_N(info));
    std::lock_guard<std::mutex> g(h->lock);
    h->idle_us = CLAMP(us, (guint64)200, (guint64)100000);
}

// The held-back frame leaves alone (lock held).  A failure cannot be the flow return of its own buffer any more: it is remembered for
// the next transform() and posted once the lock is released.
static inline void mvfx_pair_flush_locked(MvfxPairHold *h, GstObject *element, MvfxPairLaunch launch, MvfxUnrefLater *later, MvfxPostLater *post)
{
    if (!h->out_mem) return;
    GstMemory *in = h->in_mem, *out = h->out_mem;
    h->in_mem = h->out_mem = NULL;
    if (h->device >= 0 && mvfx_current_device() != h->device) mvfx_set_device(h->device); // the timer thread, an application thread
    // the marks stay until the fences are published (fence_end): a user on another thread runs into the hold's lock meanwhile
    mvfx_hip_memory_acquire_as_owner(in, h->st, element);
    mvfx_hip_memory_acquire_as_owner(out, h->st, element);
    // one fence for the launch, not one per block, and carried by the kernel itself (mvfx_hip_fence_begin / _end)
    GstMemory *const both[2] = {out, in};
    MvfxFenceScope fs;
    mvfx_hip_fence_begin(&fs, both, in ? 2 : 1, h->st, FALSE);
    const int rc = launch(element, in ? &h->fi : &h->fo, &h->fo, 1, h->st);
    mvfx_hip_fence_end(&fs, h->st, element, element);
    later->add(in);
    later->add(out);
    h->n_singles++;
    if (rc != MVFX_OK) {
        if (h->failed_rc == MVFX_OK) {
            h->failed_rc = rc;
            h->failed_text = g_strdup(mvfx_last_error());
        }
        post->set(element, rc);
    }
}

// What transform() returns when mvfx_pair_submit said MVFX_PAIR_FAILED_EARLIER: the reference would have returned this error from the
// failed frame's own call (hsvfilter/imp.rs:322-326: the panic in transform_frame_ip becomes the element's error and GST_FLOW_ERROR)
static inline GstFlowReturn mvfx_pair_flow_error(MvfxPairHold *h, GstObject *element)
{
    int rc;
    gchar *text;
    {
        std::lock_guard<std::mutex> g(h->lock);
        rc = h->failed_rc;
        text = g_strdup(h->failed_text ? h->failed_text : "");
    }
    GST_ELEMENT_ERROR(GST_ELEMENT(element), LIBRARY, FAILED, ("%s", text), ("the frame before this one failed after its transform call had "
                      "returned (pair launches): mvfx status %d (%s)", rc, mvfx_status_string(rc)));
    g_free(text);
    return GST_FLOW_ERROR;
}

static inline void mvfx_pair_flush(MvfxPairHold *h, GstObject *element, const MvfxPairOps *ops) // EOS, flush-start
{
    MvfxUnrefLater later;
    MvfxPostLater post;
    std::lock_guard<std::mutex> g(h->lock);
    mvfx_pair_flush_locked(h, element, ops->launch, &later, &post);
}

static inline void mvfx_pair_stop(MvfxPairHold *h, GstObject *element, const MvfxPairOps *ops) // stop(): the error state ends with the stream
{
    mvfx_idle_cancel(element);
    mvfx_pair_flush(h, element, ops);
    std::lock_guard<std::mutex> g(h->lock);
    h->failed_rc = MVFX_OK;
    g_free(h->failed_text);
    h->failed_text = NULL;
}

// The flush registered on the blocks: somebody looked at a held-back frame
static inline void mvfx_pair_flush_foreign(MvfxPairHold *h, GstObject *element, const MvfxPairOps *ops)
{
    MvfxUnrefLater later;
    MvfxPostLater post;
    std::lock_guard<std::mutex> g(h->lock);
    if (h->out_mem && mvfx_pair_mode() == 1 && ++h->foreign_streak >= kMvfxPairStreak) {
        h->foreign_streak = 0;
        h->direct_left = kMvfxPairDirect;
    }
    mvfx_pair_flush_locked(h, element, ops->launch, &later, &post);
}

// The flush registered with the timer thread: no second buffer within one frame interval (a live source that stalls, a paused
// application): the frame leaves alone.  Re-arming on every hold moves the deadline, so whatever is held here has waited long enough.
static inline void mvfx_pair_flush_idle(MvfxPairHold *h, GstObject *element, const MvfxPairOps *ops)
{
    MvfxUnrefLater later;
    MvfxPostLater post;
    std::lock_guard<std::mutex> g(h->lock);
    if (h->out_mem) {
        h->n_idle++;
        if (g_getenv("MVFX_ELEMENT_PAIR_STATS"))
            g_printerr("%s: idle flush of a frame held for %" G_GINT64_FORMAT " us (interval %" G_GUINT64_FORMAT " us)\n", GST_OBJECT_NAME(element),
                       g_get_monotonic_time() - h->held_since, h->idle_us);
    }
    mvfx_pair_flush_locked(h, element, ops->launch, &later, &post);
}

// One buffer of the element (inbuf == NULL: in place, `fi` is ignored).  `compatible`: the held-back frame (if any) may share a launch
// with this one as far as the element's own state goes; geometry, formats and distinct blocks are checked here.  `store`: called
// under the lock when this frame becomes the held-back one (the element copies its settings next to it).  Returns the MVFX_* code of
// a launch -- or of an EARLIER held-back frame's failed launch --, MVFX_OK when the frame was held back.
template <typename Store>
static inline int mvfx_pair_submit(MvfxPairHold *h, GstObject *element, const MvfxPairOps *ops, GstBuffer *inbuf, GstBuffer *outbuf,
                                   const mvfx_frame &fi, const mvfx_frame &fo, mvfx_stream st, gboolean compatible, Store store)
{
    GstMemory *in = inbuf ? gst_buffer_peek_memory(inbuf, 0) : NULL, *out = gst_buffer_peek_memory(outbuf, 0);
    GstMemory *const watched = in ? in : out; // what upstream writes
    // input still being written (an upstream kernel or the source's copy in flight, or held back): this element is a stage of a
    // dependent chain -- no holding back behind that (asked BEFORE the foreign flush below turns a held-back kernel into a fence)
    const gboolean chained = mvfx_pair_mode() == 1 && mvfx_hip_memory_busy(watched, element);
    // what OTHER elements hold back on the two blocks (an upstream in-place filter's kernel on our input, a former user's on a
    // recycled output block) leaves now, before our lock is taken: under the lock no foreign flush ever runs (mvfxhipmemory.cpp)
    if (in) mvfx_hip_memory_flush_foreign(in, element);
    mvfx_hip_memory_flush_foreign(out, element);
    MvfxUnrefLater later;
    MvfxPostLater post;
    std::unique_lock<std::mutex> g(h->lock);
    h->n_buffers++;
    if (h->failed_rc != MVFX_OK) // the frame before this one failed after its call had returned: this call carries the error
        return MVFX_PAIR_FAILED_EARLIER;
    if (chained || h->direct_left) { // ... or every held-back frame was flushed by somebody's look lately: a plain launch per buffer for a while
        if (h->direct_left) h->direct_left--;
        h->n_direct++;
        mvfx_pair_flush_locked(h, element, ops->launch, &later, &post);
        return h->failed_rc != MVFX_OK ? MVFX_PAIR_FAILED_EARLIER : MVFX_PAIR_NOT_TAKEN;
    }
    const auto same_shape = [](const mvfx_frame &a, const mvfx_frame &b) {
        return a.width == b.width && a.height == b.height && a.stride == b.stride && a.format == b.format;
    };
    if (h->out_mem && (!compatible || (in && !same_shape(h->fi, fi)) || !same_shape(h->fo, fo) || (in && (h->in_mem == in || h->out_mem == in)) ||
                       h->in_mem == out || h->out_mem == out))
        mvfx_pair_flush_locked(h, element, ops->launch, &later, &post); // the held-back frame goes first, alone
    if (h->failed_rc != MVFX_OK) return MVFX_PAIR_FAILED_EARLIER;
    if (!h->out_mem) {
        h->in_mem = in ? gst_memory_ref(in) : NULL;
        h->out_mem = gst_memory_ref(out);
        if (in) h->fi = fi;
        h->fo = fo;
        h->st = st;
        h->device = mvfx_current_device();
        h->held_since = g_get_monotonic_time();
        const guint64 idle_us = h->idle_us;
        store();
        g.unlock();
        // (set_deferred runs whatever somebody else holds back on the block first -- an upstream element's kernel on our input --
        // and must not be called under the lock: that flush may be ours on another block)
        if (in) mvfx_hip_memory_set_deferred(in, ops->looked_at, element);
        mvfx_hip_memory_set_deferred(out, ops->looked_at, element);
        mvfx_idle_arm(element, ops->idle, idle_us);
        // a flush from the application thread (flush-start, stop()) or the timer may have taken the frame between the unlock and the
        // marks: they would be stale then -- a reference on the element parked on a pool block, and a later look at that block flushing
        // whatever unrelated frame is held at that time (advisor r4).  Only this thread creates holds, so "still held" is one compare.
        g.lock();
        if (h->out_mem != out) {
            if (in) mvfx_hip_memory_clear_deferred(in, element);
            mvfx_hip_memory_clear_deferred(out, element);
        }
        return MVFX_OK;
    }
    GstMemory *first_in = h->in_mem, *first_out = h->out_mem;
    const mvfx_frame ins[2] = {in ? h->fi : h->fo, in ? fi : fo}, outs[2] = {h->fo, fo};
    h->in_mem = h->out_mem = NULL;
    static const int pair_stream_mode = g_getenv("MVFX_PAIR_STREAM") ? atoi(g_getenv("MVFX_PAIR_STREAM")) : 1;
    // consecutive PAIRS alternate between two streams of this thread (0: the second frame's own stream, experiment)
    const mvfx_stream pst = pair_stream_mode ? mvfx_thread_stream_n(h->pair_no++ & 1u) : st;
    mvfx_hip_memory_acquire_as_owner(first_in, pst, element);
    mvfx_hip_memory_acquire_as_owner(first_out, pst, element);
    mvfx_hip_memory_acquire_as_owner(in, pst, element);
    mvfx_hip_memory_acquire_as_owner(out, pst, element);
    // ONE fence for the blocks of both frames, carried by the kernel itself as its stop event: an event record costs the device a bubble
    GstMemory *all[4];
    guint n_all = 0;
    for (GstMemory *m : {first_out, out, first_in, in})
        if (m) all[n_all++] = m;
    MvfxFenceScope fs;
    mvfx_hip_fence_begin(&fs, all, n_all, pst, FALSE);
    const int rc = ops->launch(element, ins, outs, 2, pst);
    mvfx_hip_fence_end(&fs, pst, element, element);
    h->n_pairs++;
    h->foreign_streak = 0;
    later.add(first_in);
    later.add(first_out);
    return rc;
}

static inline void mvfx_pair_print_stats(MvfxPairHold *h, GstObject *element, const char *what)
{
    if (g_getenv("MVFX_ELEMENT_PAIR_STATS") && h->n_buffers)
        g_printerr("%s %s: %" G_GUINT64_FORMAT " device buffers = 2 x %" G_GUINT64_FORMAT " pair launches + %" G_GUINT64_FORMAT
                   " single launches + %" G_GUINT64_FORMAT " direct launches (%" G_GUINT64_FORMAT " of the single launches by the idle timer)\n",
                   what, GST_OBJECT_NAME(element), h->n_buffers, h->n_pairs, h->n_singles, h->n_direct, h->n_idle);
}

// The three callbacks + the ops table of an element type whose instance struct has a member `MvfxPairHold *hold`
#define MVFX_PAIR_DEFINE_OPS(prefix, Type, launch_fn)                                                                         \
    static void prefix##_looked_at_cb(GstObject *owner);                                                                      \
    static void prefix##_idle_cb(GstObject *owner);                                                                           \
    static const MvfxPairOps prefix##_pair_ops = {launch_fn, prefix##_looked_at_cb, prefix##_idle_cb};                        \
    static void prefix##_looked_at_cb(GstObject *owner)                                                                       \
    {                                                                                                                         \
        mvfx_pair_flush_foreign(reinterpret_cast<Type *>(owner)->hold, owner, &prefix##_pair_ops);                            \
    }                                                                                                                         \
    static void prefix##_idle_cb(GstObject *owner) { mvfx_pair_flush_idle(reinterpret_cast<Type *>(owner)->hold, owner, &prefix##_pair_ops); } \
    static void prefix##_flush_cb(GstObject *owner) /* EOS, flush-start */                                                    \
    {                                                                                                                         \
        mvfx_pair_flush(reinterpret_cast<Type *>(owner)->hold, owner, &prefix##_pair_ops);                                    \
    }
